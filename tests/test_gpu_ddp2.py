"""Two PROCESSES training data-parallel on the one GPU of the test box (torch.distributed, gloo backend on CUDA tensors: RCCL refuses two
ranks on one device, and no multi-GPU box is available to the tests).  It is the N > 1 step end to end in the real process layout -- rank-local
data, the flat-gradient all-reduce between backward and optimizer, 1/W folded into the unscale coefficient, the two-graph fallback of
capture() for a collective that cannot be captured -- with a different transport than the RCCL ring of a real node.
Checked: (1) the all-reduced buffer is the sum of the ranks' local gradients; (2) every rank applies identical updates (replicas stay
bit-identical, no parameter broadcast); (3) graph replay == eager; (4) the result equals the single-process run of the hand-averaged
gradient (the data-parallel step IS the large-batch step)."""
import os
import socket
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from oracle import train_step as ts          # test infrastructure: synthetic prompts only
    from test_gpu_model import build_step
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    B, hw, D = 2, 16, 64
    _, step, added = build_step(B, hw, D)        # same seeds on every rank: identical initial replicas, as DDP guarantees by broadcast
    step.world = world
    g = torch.Generator().manual_seed(100 + rank)  # rank-local data
    batches = []
    for _ in range(3):
        batches.append((ts.synthetic_ids(B, added, g), ts.synthetic_ids(B, added, g, prior=True), torch.randn(B, 4, hw, hw, generator=g),
                        torch.randn(B, 4, hw, hw, generator=g), torch.randint(0, 1000, (B,), generator=g)))

    def feed(i):
        ids, pids, x0, noise, t = batches[i]
        step.input_ids.copy_(ids); step.prior_ids.copy_(pids); step.x0.copy_(x0); step.noise.copy_(noise); step.timesteps.copy_(t)

    # step 0, eager, with the gradient exchange taken apart
    feed(0)
    step.draw(); step.forward_backward()
    local = step.flat_grad.clone()
    step.all_reduce()
    reduced = step.flat_grad.clone()
    step.optimizer_step()
    # steps 1 and 2 through capture(): gloo cannot be captured -> two graphs around the eager all-reduce
    feed(1)
    step.capture(warmup=0)
    mode = step.graph_mode
    step.replay()
    feed(2)
    step.replay()
    torch.cuda.synchronize()
    te = step.te
    torch.save({"local": local.cpu(), "reduced": reduced.cpu(), "mode": mode, "lora_A": te.lora_A.cpu(), "lora_B": te.lora_B.cpu(),
                "added": te.token_table[te.first_added:].cpu(), "scalars": step.scalars(), "batches": batches},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_process_data_parallel_step_on_one_gpu(tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, f"rank{r}.pt"), weights_only=False) for r in (0, 1))
    assert r0["mode"] == r1["mode"] == "two+eager-rccl"
    assert not torch.equal(r0["local"], r1["local"])                                   # rank-local data -> different local gradients
    torch.testing.assert_close(r0["reduced"], r0["local"] + r1["local"], rtol=1e-6, atol=0)  # (1)
    assert torch.equal(r0["reduced"], r1["reduced"])
    for k in ("lora_A", "lora_B", "added"):                                                 # (2), (3): after one eager + two replayed steps
        assert torch.equal(r0[k], r1[k]), k
    assert r0["scalars"]["opt_steps"] == r1["scalars"]["opt_steps"] == 3.0
    # (4) one process, same three exchanges done by hand: backward on each rank's batch, sum, grad_div = 2
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_model import build_step
    B, hw, D = 2, 16, 64
    _, step, added = build_step(B, hw, D)
    step.world = 1
    import textboost_amd.ops as ops  # noqa: F401
    for i in range(3):
        grads = []
        for r in (r0, r1):
            ids, pids, x0, noise, t = r["batches"][i]
            step.input_ids.copy_(ids); step.prior_ids.copy_(pids); step.x0.copy_(x0); step.noise.copy_(noise); step.timesteps.copy_(t)
            step.draw(); step.forward_backward()
            grads.append(step.flat_grad.clone())
        step.flat_grad.copy_(grads[0] + grads[1])
        step.world = 2          # grad_div of the optimizer tail
        step.optimizer_step()
        step.world = 1
    torch.cuda.synchronize()
    te = step.te
    torch.testing.assert_close(te.lora_A.cpu(), r0["lora_A"], rtol=0, atol=0)
    torch.testing.assert_close(te.lora_B.cpu(), r0["lora_B"], rtol=0, atol=0)
    torch.testing.assert_close(te.token_table[te.first_added:].cpu(), r0["added"], rtol=0, atol=0)
