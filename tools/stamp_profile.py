"""Diagnostic: in-kernel phase stamps (s_memtime) of tile_fwd / bwd.  GPU box only; PRISM_DBG=8."""
import os, sys, contextlib, io
os.environ["PRISM_DBG"] = os.environ.get("PRISM_DBG", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prism_amd.config import baseline_config
from prism_amd.learner import Learner
from prism_amd.synthetic import fill_replay
cfg = baseline_config(int(sys.argv[1]) if len(sys.argv) > 1 else 2, device="cuda:0")
cfg.hip_graph = False
ln = Learner()
with contextlib.redirect_stdout(io.StringIO()):
    ln.configure(cfg, obs_shape=(10, 10, 4), n_actions=6)
fill_replay(ln.experience_buffer, 20000 if cfg.experience_replay_capacity > 20000 else cfg.experience_replay_capacity, seed=0)
ag = ln.agent
for _ in range(5): ln.step(eager=True)
st = torch.zeros(4096 * 64, dtype=torch.int64, device="cuda:0")
ag._desc.dbg_stamps = st.data_ptr()
# tile_fwd and bwd share the stamp buffer: run one step, read after each kernel is impossible -> use
# the fact that bwd overwrites slots 0..4 of its own blocks; dump both by running twice with masks
def run():
    st.zero_(); ln.step(eager=True); torch.cuda.synchronize(); return st.cpu().numpy().reshape(4096, 64)
s = run()
f = s[:256, :8].astype(np.float64)
names = ["weights issue, tau, cos basis", "phi GEMM (K=64) + epilogue", "barrier", "LayerNorm(1024)", "trunk GEMM (K=1024)", "fold of the K slices", "LayerNorm(128) + head"]
print("tile_fwd (median over 256 workgroups, shader-clock ticks):")
for k in range(7):
    print(f"   {names[k]:24s} {np.median(f[:, k + 1] - f[:, k]):9.0f}")
print(f"   total {np.median(f[:, 7] - f[:, 0]):9.0f}   start spread {f[:, 0].max() - f[:, 0].min():9.0f}   end spread {f[:, 7].max() - f[:, 7].min():9.0f}")
ow = s[s[:, 33] != 0][:, [7, 32, 33]].astype(np.float64)
n_pub = int((s[:, 32] != 0).sum()) - len(ow)
if len(ow): print("   fused loss tail of the current-state tiles (barrier | loss of the tile's samples, incl. the wait for the publishers):", [int(np.median(ow[:, k + 1] - ow[:, k])) for k in range(2)], f"({len(ow)} consumer tiles, {n_pub} publisher tiles)")
nb = int((s[:, 8] != 0).sum())
b = s[:nb, 8:13].astype(np.float64)
print(f"bwd ({nb} workgroups):")
for k, name in enumerate(["consts", "tile loop", "barrier", "reduce+write"]):
    print(f"   {name:24s} {np.median(b[:, k + 1] - b[:, k]):9.0f}")
print(f"   total {np.median(b[:, 4] - b[:, 0]):9.0f}   start spread {b[:, 0].max() - b[:, 0].min():9.0f}  end spread {b[:, 4].max() - b[:, 4].min():9.0f}")
bt = s[:nb, [12, 26]].astype(np.float64)
if bt[:, 1].any(): print("   conv tail:", int(np.median(bt[:, 1] - bt[:, 0])))

B = cfg.batch_size
fr = s[:B, 27:32].astype(np.float64)
if fr[:, 0].any():
    print("front sample blocks (top+query | descent | n-step | gather+conv): med", [int(np.median(fr[:, k + 1] - fr[:, k])) for k in range(4)], "total", int(np.median(fr[:, 4] - fr[:, 0])))
ft = s[:B, [27, 24, 28]].astype(np.float64)
if ft[:, 1].any(): print("   top+query detail (loads + LDS staging + barrier | p_sum/p_min fold by lane 0):", [int(np.median(ft[:, k + 1] - ft[:, k])) for k in range(2)])
fx = s[B:B + 1024, [27, 31]].astype(np.float64)
fx = fx[fx[:, 0] != 0]
if len(fx): print(f"front extra blocks (u/v, weight packing): n={len(fx)} dur med {int(np.median(fx[:, 1] - fx[:, 0]))} max {int(np.max(fx[:, 1] - fx[:, 0]))}")
# post kernel: per-block start/end (slots 13, 14); roles by block index (conv | slab | small | ... | writeback)
n_conv = 1 if (cfg.use_iqn and not cfg.use_ids and not cfg.use_dqn) else (B + 3) // 4
pb = s[:, 13] != 0
ps_, pe_ = s[pb, 13].astype(np.float64), s[pb, 14].astype(np.float64)
idx = np.nonzero(pb)[0]
t0 = ps_.min()
print(f"post ({pb.sum()} blocks): span {pe_.max() - t0:9.0f} ticks")
def rep(name, m):
    if m.any():
        print(f"   {name:12s} n={m.sum():4d} start {np.median(ps_[m]-t0):8.0f}  dur med {np.median(pe_[m]-ps_[m]):8.0f} max {np.max(pe_[m]-ps_[m]):8.0f}  end max {np.max(pe_[m])-t0:8.0f}")
rep("conv", idx < n_conv)
rep("slab", (idx >= n_conv) & (idx < n_conv + 49))
rep("small", (idx >= n_conv + 49) & (idx < n_conv + 57))
rep("rest", (idx >= n_conv + 57) & (idx < idx.max()))
sm = s[n_conv + 49:n_conv + 57].astype(np.float64)
cv = s[:n_conv].astype(np.float64)
if n_conv > 1: print("   conv phases (partials | publish+ticket | rest): med", [int(np.median(x)) for x in (cv[:, 20] - cv[:, 13], cv[:, 21] - cv[:, 20], cv[:, 14] - cv[:, 21])], "max", [int(np.max(x)) for x in (cv[:, 20] - cv[:, 13], cv[:, 21] - cv[:, 20], cv[:, 14] - cv[:, 21])])
if n_conv > 1: print("   conv partials (stage | mac | reduce): med", [int(np.median(x)) for x in (cv[:, 22] - cv[:, 13], cv[:, 23] - cv[:, 22], cv[:, 20] - cv[:, 23])])
print("   small phases (load | D | action rounds | b1+W2 tail | end):", [int(np.median(sm[:, 16 + k] - (sm[:, 13] if k == 0 else sm[:, 15 + k]))) for k in range(4)], int(np.median(sm[:, 14] - sm[:, 19])))
rep("writeback", idx == idx.max())
# generic view: the ten longest post blocks (index, duration) -- for configurations with Q-head roles
d_ = pe_ - ps_
order = np.argsort(-d_)[:10]
print("   longest post blocks (index: ticks):", ", ".join(f"{int(idx[i])}: {int(d_[i])}" for i in order))
print("   post block duration by index decile:", [int(np.median(d_[(idx >= lo_) & (idx < hi_)])) if ((idx >= lo_) & (idx < hi_)).any() else 0
      for lo_, hi_ in zip(np.linspace(0, idx.max() + 1, 11)[:-1], np.linspace(0, idx.max() + 1, 11)[1:])])
# tile_fwd per launch-order range (passes differ in cost: IQN publisher / consumer tiles, Q-head tiles)
tb = s[:, [0, 7]].astype(np.float64)
tb = tb[tb[:, 0] != 0]
if len(tb) > 256:
    d = tb[:, 1] - tb[:, 0]
    q = len(d) // 8
    print("tile_fwd block duration by launch-order octile:", [int(np.median(d[i * q:(i + 1) * q])) for i in range(8)], f"({len(d)} tiles)")
