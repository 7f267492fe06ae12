// Internal interface between attention.hip (dispatch) and attention_il.hip (software-pipelined kernels of the hd = 40 self-attention shape).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/textboost_hip.h"

bool tb_attn_il_fwd_ok(const tb_attn_desc& d);
int tb_attn_il_fwd(const tb_attn_desc& d, hipStream_t s, int remap);
bool tb_attn_il_dkv_ok(const tb_attn_desc& d);
int tb_attn_il_dkv(const tb_attn_desc& d, hipStream_t s, int remap);
bool tb_attn_il_dq_ok(const tb_attn_desc& d);
int tb_attn_il_dq(const tb_attn_desc& d, hipStream_t s, int remap, int publish);
// attention_small.hip: the whole backward of a short sequence (Sq = Skv <= 96, hd = 64) in one launch
bool tb_attn_small_bwd_ok(const tb_attn_desc& d);
int tb_attn_small_bwd(const tb_attn_desc& d, hipStream_t s);
