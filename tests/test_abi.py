"""No GPU needed: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/prism_hip.h declares; argument checking works without touching a device."""
import ctypes
import os
import re

import pytest

from tests import helpers as H


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    from prism_amd import _native as N
    if not os.path.exists(N.LIB_PATH):
        g.build()
    return N.lib()


def declared_symbols():
    src = open(os.path.join(H.ROOT, "include", "prism_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(prism_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from prism_amd import _native as N
    names = declared_symbols()
    assert len(names) >= 20
    raw = ctypes.CDLL(N.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in prism_hip.h but not exported"
        assert n in N.SIGNATURES, f"{n} has no ctypes signature in prism_amd/_native.py"
    assert set(N.SIGNATURES) == set(names)
    assert lib.prism_abi_version() == 3
    # (documentation drift guard: DESIGN.md quotes the number of entry points)
    assert f"({len(names)} entry points)" in open(os.path.join(H.ROOT, "DESIGN.md")).read()


def test_struct_layouts_match_header():
    """ctypes mirrors must have the C sizes (guards against silent field drift)."""
    from prism_amd import _native as N
    assert ctypes.sizeof(N.ReplayDesc) == 8 + 8 + 4 + 4 + 10 * 8 + 16 * 8
    assert ctypes.sizeof(N.ModelDims) == 16 * 4 + 4 * 4 + 4          # (+ squish_fn)
    assert ctypes.sizeof(N.ParamOffsets) == 23 * 8
    assert ctypes.sizeof(N.AdamHyper) == 4 * 8 + 2 * 4
    # learner desc: dims(84 + 4 padding) off(184) batch+embed(8) 6 ptrs, 7 ptrs, 4 ptrs, seed/offset/rng (24), 6 ptrs + size_t + hyper(40), host_status
    assert ctypes.sizeof(N.LearnerDesc) == 88 + 184 + 8 + 6 * 8 + 7 * 8 + 4 * 8 + 24 + 24 + 8 + 8 + 7 * 8 + 8 + 40 + 8
    assert ctypes.sizeof(N.DirectDesc) == 8 + 8 * 8 + 8 * 8 + 8 + 8 + 8 + 8


def test_argument_checks_without_device(lib):
    from prism_amd import _native as N
    assert lib.prism_per_sample(None, 10, 4, None, 0, 0, 0.5, None, None, None) == -1
    assert b"null descriptor" in lib.prism_last_error()
    d = N.ReplayDesc()
    d.capacity, d.tree_capacity = 100, 64
    assert lib.prism_replay_init(ctypes.byref(d), None) == -1
    assert b"tree_capacity" in lib.prism_last_error()
    dims = N.ModelDims()
    assert lib.prism_learner_workspace_bytes(ctypes.byref(dims), 256) == 0          # unsupported dims
    assert lib.prism_learner_supported(ctypes.byref(dims), 256) == -3
    dims.in_channels, dims.n_actions, dims.embed_dim, dims.use_iqn = 4, 6, 1024, 1
    dims.n_basis, dims.iqn_layers, dims.iqn_width, dims.n_tau, dims.n_tau_next, dims.use_layer_norm = 64, 1, 128, 8, 8, 1
    assert lib.prism_learner_supported(ctypes.byref(dims), 256) == 0
    assert lib.prism_learner_workspace_bytes(ctypes.byref(dims), 256) > 8 * 2 ** 20
    assert lib.prism_learner_supported(ctypes.byref(dims), 255) == -3                # B*T not a multiple of 16


def test_product_never_imports_oracle_and_has_no_cpu_path():
    import subprocess
    import sys
    root = os.path.join(H.ROOT, "prism_amd")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("# oracle", ""), f"{f} references the oracle"
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from prism_amd.config import baseline_config\n"
            "from prism_amd.factory import exp_buffer_factory\n"
            "from prism_amd._native import NativeLibraryError\n"
            "try:\n    exp_buffer_factory.build_exp_buffer(baseline_config(2, device='cpu'))\n"
            "except NativeLibraryError as e:\n    print('LOUD', e)\n" % H.ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert "LOUD" in out.stdout, out.stdout + out.stderr


@pytest.fixture(scope="module")
def device_isa(tmp_path_factory):
    """gfx950 assembly of the two translation units that hold every kernel: {source name: text}."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    ROOT = H.ROOT
    csrc = os.path.join(ROOT, "prism_amd", "csrc")
    tmp = tmp_path_factory.mktemp("isa")
    out = {}
    for src in ("learner.hip", "replay.hip"):
        o = tmp / (src + ".s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                        "-I" + os.path.join(ROOT, "include"), "-I" + csrc, "-S", "--cuda-device-only",
                        "-o", str(o), os.path.join(csrc, src)], check=True, timeout=900,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out[src] = o.read_text()
    return out


def test_kernels_use_no_scratch_memory(device_isa):
    """A register array that lands in scratch (private segment) costs a kernel several microseconds per
    launch on this path (it happened three times during development: indexed selects, conditionally
    initialised arrays, kernel-argument structs passed by reference).  Require a zero private segment
    for every kernel."""
    bad = []
    for src, text in device_isa.items():
        for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
            seg = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", m.group(2))
            uses = re.search(r"scratch_(load|store)", text[text.find(m.group(1) + ":"):text.find(".amdhsa_kernel " + m.group(1))])
            if seg and int(seg.group(1)) > 0 and uses:
                bad.append((m.group(1), int(seg.group(1))))
    assert not bad, f"kernels with scratch traffic: {bad}"


def test_release_tickets_drain_every_wave_first(device_isa):
    """Cross-workgroup hand-offs (the conv-partial ticket, the grid barrier of the fused tail) publish data other
    workgroups read once the ticket says so.  The protocol: EVERY wave drains its own stores (s_waitcnt vmcnt(0)), the
    workgroup meets (s_barrier), then one lane draws the ticket (a returning global_atomic_add) -- after writing the L2 back
    (buffer_wbl2) where the data went out through ordinary stores, without where every store was written through (far_store:
    the conv ticket since round 4, the light arrivals of the grid barrier).  A barrier alone only proves the stores were
    issued; hipcc moves or drops the wait unless it is pinned.  Check the emitted ISA: walking back from every ticket that
    has a workgroup barrier in front of it, the last s_waitcnt before that s_barrier waits for vmcnt(0), with no store in
    between."""
    text = device_isa["learner.hip"]
    lines = text.splitlines()
    tickets = [i for i, l in enumerate(lines) if re.search(r"\bglobal_atomic_add(_x2)?\b.*\bsc0\b", l)]
    checked, released, bad = 0, 0, []
    for i in tickets:
        j, wb = i - 1, False
        while j >= 0 and "s_barrier" not in lines[j] and not re.match(r"^\S+:\s*$", lines[j].split(";")[0]) or \
                (j >= 0 and lines[j].startswith(".LBB")):
            wb = wb or "buffer_wbl2" in lines[j]
            j -= 1
        if j < 0 or "s_barrier" not in lines[j]:
            continue          # no workgroup barrier in front of it: a lane's own counter, nothing of the other waves is handed over
        f = i
        while f > 0 and not re.match(r"^_ZN\S+:", lines[f]):
            f -= 1
        if not wb and "iqn_post_kernel" not in lines[f]:
            continue          # a plain counter outside the post launch ("last workgroup advances the step count"): nothing is handed over
        checked += 1
        released += int(wb)
        k, ok = j - 1, False
        while k >= 0:                     # back from the barrier: a vmcnt(0) wait before any memory instruction or label
            ins = lines[k].split(";")[0].strip()
            if ins.startswith("s_waitcnt") and "vmcnt(0)" in ins:
                ok = True
                break
            if re.match(r"(global|buffer|flat|scratch)_", ins) or "s_barrier" in ins or re.match(r"^\S+:$", ins):
                break
            k -= 1
        if not ok:
            bad.append((i + 1, lines[i].strip()))
    assert released >= 2, f"expected the full-release arrivals of the two fused-tail instantiations, found {released}"
    assert checked >= 7, f"expected the conv tickets of five post-kernel forms and the grid barriers, found {checked}"
    assert not bad, f"release tickets whose workgroup barrier is not preceded by a vmcnt(0) drain: {bad}"
