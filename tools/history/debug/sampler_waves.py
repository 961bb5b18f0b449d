"""Per-wave entry / exit stamps of sample_phase2_kernel (inline reset): lifetimes, start spread, waves per SIMD."""
import os, sys, collections
os.environ["GX_SAMPLER_STAMPS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from guardx_amd import _native
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
env = bench.make_engine(2000, 0, 1, n_candidates=M)
env.set_prefetch(-1)
for _ in range(3): env.reset()
lib = _native.load()
st = torch.zeros(65536 + 4 * 16384, dtype=torch.int64, device=dev)
_native.check(lib.gx_debug_stamps(env._h, st.data_ptr()))
env.reset(); torch.cuda.synchronize()
_native.check(lib.gx_debug_stamps(env._h, None))
s = st.cpu().numpy()[65536:].reshape(-1, 4)
s = s[s[:, 0] != 0]
life = s[:, 1] - s[:, 0]
print(f"M={M}: {len(s)} active waves (s_memtime ticks; the time base is per XCD)")
hw, xcc = s[:, 2], s[:, 3] & 0xf
simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = list(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist(), simd.tolist()))
per_simd = collections.Counter(key)
per_cu = collections.Counter(k[:4] for k in key)
print("  distinct SIMDs", len(per_simd), "distinct CUs", len(per_cu))
print("  waves per SIMD histogram:", sorted(collections.Counter(per_simd.values()).items()))
print("  waves per CU histogram:", sorted(collections.Counter(per_cu.values()).items()))
# lifetime against the number of waves on the same SIMD
by = collections.defaultdict(list)
for k, l in zip(key, life): by[per_simd[k]].append(l)
for n in sorted(by): print(f"  SIMDs with {n} waves: median lifetime {np.median(by[n]):.0f}")
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    t0 = s[m, 0].min(); st_ = s[m, 0] - t0; en = s[m, 1] - t0
    print(f"  XCD {x}: {m.sum()} waves; start offsets median {np.median(st_):.0f} p90 {np.percentile(st_,90):.0f} max {st_.max()};"
          f" end offsets median {np.median(en):.0f} p90 {np.percentile(en,90):.0f} max {en.max()}; lifetime median {np.median(life[m]):.0f} max {life[m].max()}")
m = xcc == 0
t0 = s[m, 0].min()
order = np.argsort(s[m, 0])
print("  XCD 0, every 40th wave by start: (start, lifetime, waves on its SIMD)")
kk = [key[i] for i in np.nonzero(m)[0]]
for i in order[::40]: print("    ", s[m][i, 0] - t0, life[m][i], per_simd[kk[i]])
