import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda", 0)
for N in (2000, 64):
    env = bench.make_engine(N, 0, 1, n_candidates=200000)
    tape = bench.action_tape(200, N, 0, dev)
    env.reset()
    for mode in ("step+rd", "step only"):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for rep in range(10):
            for t in range(200):
                env.step(tape[t])
                if mode == "step+rd": env.reset_done()
        t1 = time.perf_counter()      # host time to enqueue
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"N={N} {mode}: host enqueue {(t1-t0)/2000*1e6:.2f} us/step, total {(t2-t0)/2000*1e6:.2f} us/step")
    # pre-sliced actions (no tape[t] indexing cost)
    acts = [tape[t] for t in range(200)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for rep in range(10):
        for a in acts:
            env.step(a); env.reset_done()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"N={N} step+rd, pre-sliced actions: host {(t1-t0)/2000*1e6:.2f}, total {(t2-t0)/2000*1e6:.2f} us/step")
    env.close()

# native call alone (prebuilt ctypes arguments): how much of the per-step host time is the HIP launch path
import ctypes as C
from guardx_amd import _native
env = bench.make_engine(2000, 0, 1, n_candidates=200000)
env.reset()
tape = bench.action_tape(200, 2000, 0, dev)
o, r, d, info = env.step(tape[0])
lib = _native.load()
slot = env._slab[0]
p = slot[6]
aptr = C.c_void_p(tape[1].data_ptr())
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
spec = C.c_int32(0); sref = C.byref(spec)
f = lib.gx_step_rd; h = env._h
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5000):
    f(h, aptr, p[0], p[1], p[2], p[3], p[4], p[5], sref, stream)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"gx_step_rd alone: host {(t1-t0)/5000*1e6:.2f} us/call, total {(t2-t0)/5000*1e6:.2f}")
g = lib.gx_reset_done_commit
t0 = time.perf_counter()
for _ in range(5000):
    g(h)
print(f"gx_reset_done_commit alone: {(time.perf_counter()-t0)/5000*1e6:.2f} us/call")
