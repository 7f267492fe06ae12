#!/bin/bash
# build_variant.sh <suffix> "<cflags>"  -> textboost_amd/libtextboost_hip<suffix>.so (fp16 build only)
TB_SKIP_BF16=1 TB_LIB_SUFFIX=$1 TB_CFLAGS="$2" python -m textboost_amd.build 2>&1 | tail -1
