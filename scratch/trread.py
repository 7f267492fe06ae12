import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "trread.so"))
def run(addr):
    a = torch.tensor(addr, dtype=torch.int32, device="cuda"); out = torch.zeros(256, device="cuda")
    lib.trread(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return out.view(64, 4).int().tolist()
# experiment 1: lane l reads 8 bytes at element offset 4*l  (contiguous): what does each lane get?
r = run([8 * l for l in range(64)])
print("exp1 addr = 8*l bytes (elements 4l..4l+3):")
for l in range(0, 64, 1): print(l, r[l], end=" | " if l % 4 != 3 else "\n")
# experiment 2: row-major matrix with row stride 64 elements: lane l points at row (l % 16), col 4*(l // 16)
r = run([2 * ((l % 16) * 64 + 4 * (l // 16)) for l in range(64)])
print("exp2 lane -> row l%16, col 4*(l//16) of a [*,64] matrix (value = row*64 + col):")
for l in range(64): print(l, [(x // 64, x % 64) for x in r[l]], end=" | " if l % 2 != 1 else "\n")
