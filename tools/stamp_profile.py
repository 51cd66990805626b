"""Diagnostic: in-kernel phase stamps (s_memtime) of tile_fwd / bwd.  GPU box only; PRISM_DBG=8."""
import os, sys, contextlib, io
os.environ["PRISM_DBG"] = os.environ.get("PRISM_DBG", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prism_amd.config import baseline_config
from prism_amd.learner import Learner
from prism_amd.synthetic import fill_replay
_arg = sys.argv[1] if len(sys.argv) > 1 else "2"
if _arg in ("add", "add_per", "add_ids", "sub"):
    # the reference's ablation presets (tools/bench_presets.py): width 256, T = 32, batch 64
    from prism_amd import config as C
    _base, _over = {"add": (C.ADDITIVE_ABLATION_BASE_CONFIG, {}),
                    "add_per": (C.ADDITIVE_ABLATION_BASE_CONFIG, dict(use_per=True, n_step_returns_length=3, use_layer_norm=True)),
                    "add_ids": (C.ADDITIVE_ABLATION_BASE_CONFIG, dict(use_ids=True, ids_n_q_head_model_layers=2, ids_n_q_heads=10,
                                                                      ids_q_head_feature_dim=256, ids_ensemble_variation_coef=0)),
                    "sub": (C.SUBTRACTIVE_ABLATION_BASE_CONFIG, {})}[_arg]
    cfg = C.derive(_base, device="cuda:0", experience_replay_capacity=100_000, log_to_wandb=False, **_over)
else:
    cfg = baseline_config(int(_arg), device="cuda:0")
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
nfw = int((s[:, 0] != 0).sum())
f = s[:nfw, :7].astype(np.float64)
names = ["first requests, tau, cos basis", "streamed phi + trunk products", "partials out + barrier", "fold, LayerNorm, head",
         "loss of the tile's samples", "head/LN backward + per-sample sums"]
print(f"fwd_tile ({nfw} workgroups; median shader-clock ticks):")
for k in range(6):
    d = f[:, k + 1] - f[:, k]
    d = d[f[:, k + 1] != 0]
    if len(d): print(f"   {names[k]:36s} {np.median(d):9.0f}   (min {d.min():7.0f} max {d.max():7.0f})")
last = np.where(f[:, 6] != 0, f[:, 6], f[:, 4])
dur = last - f[:, 0]
def rt_spans(start, end):
    """slots 32 + k: the chip-wide 100 MHz real-time counter at stamp k (10 ns ticks) -> microseconds"""
    return (start.max() - start.min()) / 100.0, (end.max() - start.min()) / 100.0
rl = np.where(s[:nfw, 38] != 0, s[:nfw, 38], s[:nfw, 36]).astype(np.float64)
ss, sp = rt_spans(s[:nfw, 32].astype(np.float64), rl)
print(f"   workgroup total: median {np.median(dur):9.0f}  p95 {np.percentile(dur, 95):9.0f}  max {dur.max():9.0f};  start spread {ss:6.2f} us   "
      f"first start -> last end {sp:6.2f} us")
if nfw > 256:
    r0_, r1_ = s[:nfw, 32].astype(np.float64), np.where(s[:nfw, 38] != 0, s[:nfw, 38], s[:nfw, 36]).astype(np.float64)
    t00 = r0_.min()
    order = np.argsort(r0_)
    print("   tile start / end (us, real time) by start order, every 32nd:", [(int(i), round((r0_[i] - t00) / 100, 1), round((r1_[i] - t00) / 100, 1)) for i in order[::32]])
    for lo_, hi_, nm in ((0, 256, "tiles 0..255"), (256, nfw, f"tiles 256..{nfw - 1}")):
        ph = [np.median((f[lo_:hi_, k + 1] - f[lo_:hi_, k])[f[lo_:hi_, k + 1] != 0]) if (f[lo_:hi_, k + 1] != 0).any() else 0 for k in range(6)]
        rs0 = s[lo_:hi_, 32].astype(np.float64)
        print(f"   {nm}: phases {[int(x) for x in ph]} total med {int(np.median(dur[lo_:hi_]))}; starts {(rs0.min() - s[:nfw, 32].min()) / 100:.2f}..{(rs0.max() - s[:nfw, 32].min()) / 100:.2f} us")
nb = int((s[:, 8] != 0).sum())
b = s[:nb, 8:13].astype(np.float64)
print(f"bwd ({nb} workgroups):")
for k, name in enumerate(["consts", "tile loop", "barrier", "reduce+write"]):
    print(f"   {name:24s} {np.median(b[:, k + 1] - b[:, k]):9.0f}")
be = s[64:nb, [13, 8]].astype(np.float64)
be = be[be[:, 0] != 0]
if len(be): print(f"   entry -> prologue loads issued (workgroups 64..): {np.median(be[:, 1] - be[:, 0]):9.0f}")
bend = np.where(s[:nb, 26] != 0, s[:nb, 26], s[:nb, 12]).astype(np.float64)
bd = bend - b[:, 0]
rl = np.where(s[:nb, 58] != 0, s[:nb, 58], s[:nb, 44]).astype(np.float64)
ss, sp = rt_spans(s[:nb, 40].astype(np.float64), rl)
print(f"   workgroup total: median {np.median(bd):9.0f}  p95 {np.percentile(bd, 95):9.0f}  max {bd.max():9.0f};  start spread {ss:6.2f} us   "
      f"first start -> last end {sp:6.2f} us")
b2 = s[2048:2048 + nb].astype(np.float64)
if b2[:, 13].any():
    e0 = s[:nb, 8].astype(np.float64)
    print("   bwd3 prologue (ticks after the entry stamp): compute W1 split done %d, rows requested %d, past barrier %d | helper: start %d, block 0 arrived %d, staged %d, past barrier %d"
          % tuple(int(np.median(b2[:, k] - e0)) for k in (13, 14, 15, 9, 10, 11, 12)))
if b2[:, 17].any():
    print("   bwd3 block 3 (ticks): compute work %d, wait at the barrier %d | helper work %d, wait %d"
          % tuple(int(np.median(b2[:, x] - b2[:, y])) for x, y in ((17, 16), (18, 17), (20, 19), (21, 20))))
bt = s[:nb, [12, 26]].astype(np.float64)
if bt[:, 1].any(): print("   conv tail:", int(np.median(bt[:, 1] - bt[:, 0])))

B = cfg.batch_size
fr = s[:B, 27:32].astype(np.float64)
if fr[:, 0].any():
    print("front sample blocks (top+query | descent | n-step | gather+conv): med", [int(np.median(fr[:, k + 1] - fr[:, k])) for k in range(4)], "total", int(np.median(fr[:, 4] - fr[:, 0])))
ft = s[:B, [27, 24, 28]].astype(np.float64)
if ft[:, 1].any(): print("   top+query detail (loads + LDS staging + barrier | p_sum/p_min fold by lane 0):", [int(np.median(ft[:, k + 1] - ft[:, k])) for k in range(2)])
rs_, re_ = s[:B, 32 + 27].astype(np.float64), s[:B, 32 + 31].astype(np.float64)
allb = s[:, 32 + 27] != 0
ra_, rb_ = s[allb, 32 + 27].astype(np.float64), s[allb, 32 + 31].astype(np.float64)
if allb.any():
    print(f"   front real time: sample blocks start spread {(rs_.max() - rs_.min()) / 100:.2f} us, first start -> last sample end {(re_.max() - rs_.min()) / 100:.2f} us; "
          f"all {int(allb.sum())} blocks: first start -> last end {(rb_.max() - ra_.min()) / 100:.2f} us, last start at {(ra_.max() - ra_.min()) / 100:.2f} us")
    print("   sample block real-time duration: med %.2f us p95 %.2f max %.2f" % tuple(np.percentile((re_ - rs_) / 100, [50, 95, 100])))
f2 = s[2048:2048 + B].astype(np.float64)
if f2[:, 0].any():
    seq = [(s[:B, 28].astype(np.float64), "fold done"), (f2[:, 0], "mass"), (f2[:, 1], "descent"), (s[:B, 29].astype(np.float64), "weight+record"),
           (f2[:, 2], "syncthreads"), (f2[:, 3], "trip A arrived"), (f2[:, 4], "obs in LDS + barrier"), (f2[:, 5], "conv cur"), (f2[:, 6], "trip B wait"),
           (s[:B, 30].astype(np.float64), "walk + outputs"), (s[:B, 31].astype(np.float64), "barrier + conv next")]
    print("   front detail:", ", ".join(f"{n} {int(np.median(y - x))}" for (x, _), (y, n) in zip(seq[:-1], seq[1:])))
fx = s[B:B + 1024, [27, 31]].astype(np.float64)
fx = fx[fx[:, 0] != 0]
if len(fx): print(f"front extra blocks (u/v, weight packing): n={len(fx)} dur med {int(np.median(fx[:, 1] - fx[:, 0]))} max {int(np.max(fx[:, 1] - fx[:, 0]))}")
# post kernel: per-block start/end (slots 13, 14); roles by block index (conv | slab | small | ... | writeback)
# conv role of the post launch: one fold block where the backward kernel produced the partial rows (IQN-only, width 128), else one
# block per eight samples (C = 4 here: step_kernels.h post_conv_blocks)
n_conv = 1 if (cfg.use_iqn and not cfg.use_ids and not cfg.use_dqn and cfg.iqn_quantile_model_feature_dim == 128) else (B + 7) // 8
pb = s[:, 13] != 0
ps_, pe_ = s[pb, 13].astype(np.float64), s[pb, 14].astype(np.float64)
idx = np.nonzero(pb)[0]
t0 = ps_.min()
print(f"post ({pb.sum()} blocks): span {pe_.max() - t0:9.0f} ticks")
def rep(name, m):
    if m.any():
        print(f"   {name:12s} n={m.sum():4d} start {np.median(ps_[m]-t0):8.0f}  dur med {np.median(pe_[m]-ps_[m]):8.0f} max {np.max(pe_[m]-ps_[m]):8.0f}  end max {np.max(pe_[m])-t0:8.0f}")
# IQN-only layout: conv | slab sums | 8 small-tensor blocks (16 hidden units each) | writeback
n_small = cfg.iqn_hidden_layer_width // 16 if hasattr(cfg, "iqn_hidden_layer_width") else 8
n_slab = int(idx.max()) - n_conv - n_small
rep("conv", idx < n_conv)
rep("slab", (idx >= n_conv) & (idx < n_conv + n_slab))
rep("small", (idx >= n_conv + n_slab) & (idx < n_conv + n_slab + n_small))
sm = s[n_conv + n_slab:n_conv + n_slab + n_small].astype(np.float64)
cv = s[:n_conv].astype(np.float64)
if n_conv == 1 and cv[0, 20]: print("   conv fold block (entry -> role | fold loads + sums | norm partial + arrival):", [int(x) for x in (cv[0, 20] - cv[0, 13], cv[0, 21] - cv[0, 20], cv[0, 14] - cv[0, 21])])
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
sl = s[n_conv:n_conv + n_slab].astype(np.float64)
if sl[:, 7].any():
    print("   slab phases (kernel entry -> role code reached | 8 chunk loads arrived | store + norm partial written):",
          [int(np.median(x)) for x in (sl[:, 8] - sl[:, 13], sl[:, 7] - sl[:, 8], sl[:, 14] - sl[:, 7])])
# explicit role layout (STAMP_ROLES="conv:64,slab:49,small:8,qslab:32,qsmall:80,wb:1"): per role, when its blocks start, finish
# their role (slot 14) and -- fused tail -- get past the barrier (25) and end (9), ticks after the launch's first stamp
if os.environ.get("STAMP_ROLES"):
    # (the chip-wide 100 MHz real-time counter, slots 32 + k: the shader clocks of different XCDs do not compare)
    lo_ = 0
    allrows = np.arange(0, sum(int(p_.split(":")[1]) for p_ in os.environ["STAMP_ROLES"].split(",")))
    t0r = s[allrows, 32 + 13].astype(np.float64).min()
    _rot, _nb = int(os.environ.get("STAMP_ROT", "0")), len(allrows)
    print("   role            n   start med/max us   role end med/max us   past barrier med/max us   end med/max us")
    for part in os.environ["STAMP_ROLES"].split(","):
        nm, cnt = part.split(":"); cnt = int(cnt)
        rows = np.arange(lo_, lo_ + cnt); lo_ += cnt
        if _rot:      # (split form: the launch's first STAMP_ROT workgroups take the LAST roles, step_kernels.h) role -> workgroup index
            rows = np.where(rows >= _nb - _rot, rows - (_nb - _rot), rows + _rot)
        g_ = lambda k: (s[rows, 32 + k].astype(np.float64) - t0r) / 100.0
        f_ = lambda x: f"{np.median(x):7.2f}/{np.max(x):7.2f}"
        print(f"   {nm:10s} {cnt:4d}   {f_(g_(13))}   {f_(g_(14))}   {f_(g_(25)) if (s[rows, 32 + 25] != 0).any() else '-':>15s}   {f_(g_(9)) if (s[rows, 32 + 9] != 0).any() else '-':>15s}")
        late = rows[np.argsort(-g_(14))[:3]]
        print("        latest three (block: start, stamps 22 / 23 / 20 / 21, role end):",
              "; ".join(f"{int(r_)}: " + " ".join(f"{(float(s[r_, 32 + k]) - t0r) / 100.0:6.2f}" if s[r_, 32 + k] else "     -" for k in (13, 22, 23, 20, 21, 14)) for r_ in late))
# fused tail (slots 25 = past the grid barrier, 9 = Adam done), role blocks only
tb = pb & (s[:, 25] != 0)
if tb.any():
    w_, a_ = (s[tb, 25] - s[tb, 14]).astype(np.float64), (s[tb, 9] - s[tb, 25]).astype(np.float64)
    c_ = (s[tb, 15] - s[tb, 25]).astype(np.float64)
    print(f"   tail: wait at the grid barrier med {np.median(w_):.0f} min {w_.min():.0f} max {w_.max():.0f};  clip + Adam med {np.median(a_):.0f} max {a_.max():.0f}"
          f" (of which norm partials + fold: {np.median(c_):.0f})")
    rs, re_ = s[pb, 32 + 13].astype(np.float64), np.where(s[pb, 32 + 9] != 0, s[pb, 32 + 9], s[pb, 32 + 14]).astype(np.float64)
    print(f"   tail real time: first start -> last role block past the barrier {(s[tb, 32 + 25].max() - rs.min()) / 100:.2f} us, -> last end {(re_.max() - rs.min()) / 100:.2f} us;"
          f"  writeback block {(s[idx.max(), 32 + 14] - s[idx.max(), 32 + 13]) / 100:.2f} us")
