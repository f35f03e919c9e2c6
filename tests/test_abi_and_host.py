"""CPU suite: the C-ABI library loads and exports every symbol include/hriemo.h declares with the
argument list the ctypes binding uses (no compute calls without a GPU); host-side logic."""
import ctypes
import os
import re

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_decls():
    hdr = open(os.path.join(REPO, "include", "hriemo.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return re.findall(r"(\w[\w\s\*]*?)\s+(hriemo_\w+)\s*\(([^)]*)\)\s*;", hdr)


def _code(t):
    t = t.strip()
    if "*" in t or "hriemo_stream_t" in t:
        return "p"
    if t.startswith("unsigned long long"):
        return "Q"
    if t.startswith("unsigned"):
        return "I"
    if t.startswith("long"):
        return "l"
    if t.startswith("float"):
        return "f"
    if t.startswith("int"):
        return "i"
    raise ValueError(t)


def test_library_exports_every_declared_symbol_with_matching_signature():
    import hri_emo_amd  # noqa: F401
    from hri_emo_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build first: make -C hri-emo_amd/csrc (or __graft_entry__.build())"
    L = ctypes.CDLL(_lib.LIB_PATH)
    decls = _header_decls()
    assert len(decls) >= 30
    for _ret, name, args in decls:
        assert hasattr(L, name), f"{name} declared in include/hriemo.h but not exported"
        codes = "".join(_code(a) for a in args.split(",") if a.strip() and a.strip() != "void")
        if name in _lib._SIGS:
            assert _lib._SIGS[name][0] == codes, (name, codes, _lib._SIGS[name][0])
    bound = set(_lib._SIGS) | {"hriemo_last_error", "hriemo_prof_name"}
    assert {d[1] for d in decls} <= bound, {d[1] for d in decls} - bound
    assert _lib.lib().hriemo_abi_version() == 1


def test_product_path_refuses_cpu_tensors_loudly():
    import hri_emo_amd as H
    m = H.FusionWithEmotionDecoder(d_model=128, num_emotions=4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.randn(2, 8, 128), torch.randn(2, 4, 128))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(REPO, "hri-emo_amd")
    for root, _d, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("hri_emo_oracle", "oracle") or f == "__init__.py" and False or \
                    not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), (root, f)


def test_drop_in_surface_matches_reference_names_and_defaults():
    import inspect
    import hri_emo_amd as H
    from oracle import hri_emo_oracle as O
    for cls in ("CrossModalBlock", "CrossModalTransformer", "BetaGate", "EmotionDecoder", "FusionWithEmotionDecoder"):
        a, b = getattr(H, cls), getattr(O, cls)
        sa, sb = inspect.signature(a.__init__), inspect.signature(b.__init__)
        assert [(p.name, p.default) for p in sa.parameters.values()] == [(p.name, p.default) for p in sb.parameters.values()], cls
        fa, fb = inspect.signature(a.forward), inspect.signature(b.forward)
        assert list(fa.parameters) == list(fb.parameters), cls
    m = H.FusionWithEmotionDecoder(d_model=128, num_emotions=4)
    o = O.FusionWithEmotionDecoder(d_model=128, num_emotions=4)
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == [(k, tuple(v.shape)) for k, v in o.state_dict().items()]
    with pytest.raises(ValueError):
        m._ensure_3d(torch.zeros(2, 2, 2, 2))
    fm = m._build_fused_mask(torch.zeros(2, 5, dtype=torch.bool), torch.ones(2, 3, dtype=torch.bool), 4)
    assert fm.shape == (2, 4) and fm[:, :3].all() and fm[:, 3].all()      # short mask padded with PAD=True


def test_hash_rng_replica_statistics():
    import sys
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import hashrng
    m = hashrng.rows_mask(12345, 7, 512, 768, 0.1)
    assert abs(1.0 - m.mean() - 0.1) < 5e-3
    m2 = hashrng.rows_mask(12346, 7, 512, 768, 0.1)
    assert (m != m2).mean() > 0.1
    a = hashrng.attn_mask(99, 3, 2, 2, 64, 64, 0.25, b_offset=0)
    b = hashrng.attn_mask(99, 3, 1, 2, 64, 64, 0.25, b_offset=1)
    assert (a[1] == b[0]).all()          # sharding-invariant: keyed on the global utterance index


def test_collate_matches_reference_fixture_and_trim_and_bucketing():
    """hri_emo_amd.data: the trainer's collate against outputs recorded from the reference's own collate_seq_batch
    (tests/golden/collate.npz), trimming of all-PAD columns, and the length-bucketed sampler's invariants."""
    import torch
    from conftest import load_golden
    from hri_emo_amd import data
    g = load_golden("collate")
    batch = [(g[f"xa{i}"], g[f"ka{i}"], g[f"xt{i}"], g[f"kt{i}"], g[f"y{i}"]) for i in range(4)]
    h_a, m_a, h_t, m_t, labels = data.collate_seq_batch(batch, "multi_label")
    for got, key in [(h_a, "h_a"), (m_a, "mask_a"), (h_t, "h_t"), (m_t, "mask_t"), (labels, "labels")]:
        assert got.dtype == g[key].dtype and torch.equal(got, g[key]), key
    single = data.collate_seq_batch([(b[0], b[1], b[2], b[3], i % 4) for i, b in enumerate(batch)], "single_label")[4]
    assert single.dtype == torch.long and torch.equal(single, g["single"])
    # pad_to: one shape for every batch (captured steps); the reference's batch is its top-left corner, the rest is PAD
    La, Lt = h_a.shape[1], h_t.shape[1]
    fa, fma, ft, fmt, _ = data.collate_seq_batch(batch, "multi_label", pad_to=(La + 5, Lt + 3))
    assert fa.shape[1] == La + 5 and ft.shape[1] == Lt + 3 and torch.equal(fa[:, :La], h_a) and torch.equal(fmt[:, :Lt], m_t)
    assert bool(fma[:, La:].all()) and bool(fmt[:, Lt:].all()) and float(fa[:, La:].abs().sum()) == 0.0
    import pytest
    with pytest.raises(ValueError, match="pad_to"):
        data.collate_seq_batch(batch, "multi_label", pad_to=(La - 1, Lt))
    # stored masks carry PAD tails: audio valid extents 7,14,2,8 of 9,14,6,11 -> 14 stays; text 5,2,6,1 of 5,3,7,4 -> 6
    ta, tma, tt, tmt = data.trim_padding(h_a, m_a, h_t, m_t)
    assert ta.shape[1] == 14 and tt.shape[1] == 6 and torch.equal(ta, h_a[:, :14]) and torch.equal(tmt, m_t[:, :6])
    assert bool(m_t[:, 6:].all()) and bool(m_a[:, 14:].all())
    # audio is never cut below the text length (BetaGate slices h_a[:, :L_t])
    ha2, ma2 = torch.zeros(2, 10, 4), torch.ones(2, 10, dtype=torch.bool); ma2[:, :3] = False
    ht2, mt2 = torch.zeros(2, 8, 4), torch.zeros(2, 8, dtype=torch.bool)
    a3, _, t3, _ = data.trim_padding(ha2, ma2, ht2, mt2)
    assert a3.shape[1] == 8 and t3.shape[1] == 8
    obj = {"hidden": torch.ones(5, 4, dtype=torch.float16), "attention_mask": torch.tensor([1, 1, 1, 0, 0])}
    h, m = data.load_seq_feat(obj)
    assert h.dtype == torch.float32 and m.tolist() == [False, False, False, True, True]
    lengths = [int(x) for x in torch.randint(5, 400, (1003,), generator=torch.Generator().manual_seed(3))]
    batches = data.length_bucketed_batches(lengths, 32, generator=torch.Generator().manual_seed(4))
    flat = sorted(i for b in batches for i in b)
    assert flat == list(range(1003)) and all(len(b) <= 32 for b in batches)
    waste = lambda bs: sum(max(lengths[i] for i in b) * len(b) - sum(lengths[i] for i in b) for b in bs)
    plain = [list(range(s, min(s + 32, 1003))) for s in range(0, 1003, 32)]
    assert waste(batches) < 0.25 * waste(plain)


def test_packed_sequence_plan_and_row_round_trip():
    """host logic of the packed (varlen) path, no GPU needed: prefix masks give cu_seqlens and the packed-row -> padded-row index;
    anything else (a hole inside the prefix, an all-PAD row) gives no plan; pack / unpack are inverse on the valid rows"""
    import torch
    from hri_emo_amd import _ops
    B, L, d = 4, 7, 3
    lens = torch.tensor([7, 1, 4, 5])
    mask = torch.arange(L)[None, :] >= lens[:, None]
    plan = _ops.seq_plan(mask, B, L)
    assert plan is not None and plan.N == int(lens.sum()) and plan.Lmax == 7 and plan.L == L and plan.B == B
    assert plan.cu.tolist() == [0, 7, 8, 12, 17]
    assert plan.idx.tolist() == [b * L + l for b in range(B) for l in range(int(lens[b]))]
    x = torch.arange(B * L * d, dtype=torch.float32).view(B, L, d)
    packed = _ops.pack_rows(x, plan)
    assert packed.shape == (1, plan.N, d)
    back = _ops.unpack_rows(packed, plan)
    valid = ~mask
    assert torch.equal(back[valid], x[valid]) and float(back[mask].abs().sum()) == 0.0
    hole = mask.clone(); hole[0, 2] = True
    assert _ops.seq_plan(hole, B, L) is None
    allpad = mask.clone(); allpad[1, :] = True
    assert _ops.seq_plan(allpad, B, L) is None
    assert _ops.seq_plan(None, B, L) is None



def test_nested_stream_fork_is_refused_during_capture():
    """ROCm 7.2 segfaults in hipStreamEndCapture when a helper stream is forked from an already forked stream (round 2,
    gpurun_out/seg.log); _ops.fork turns that shape into a Python exception.  Stream stand-ins: only identity and wait_stream
    are used."""
    from hri_emo_amd import _ops

    class S:
        def __init__(self): self.waited = []
        def wait_stream(self, other): self.waited.append(other)

    origin, side, helper = S(), S(), S()
    _ops.fork(helper, side)                       # outside a capture anything goes
    _ops.CTX.capturing, _ops.CTX.capture_origin = True, origin
    try:
        _ops.fork(side, origin)                   # fork off the capture stream: fine
        _ops.fork(origin, side)                   # join back: fine
        with pytest.raises(RuntimeError, match="nested forks"):
            _ops.fork(helper, side)
        assert helper.waited == [side]            # only the eager call above reached the stream
    finally:
        _ops.CTX.capturing, _ops.CTX.capture_origin = False, None


def test_hook_predicates_are_per_instance():
    """one GradBuckets must not overwrite another's 'gradient hooks active' predicate"""
    from hri_emo_amd import _ops
    a = _ops.register_hook_predicate(lambda: True)
    b = _ops.register_hook_predicate(lambda: False)
    assert _ops.grad_hooks_active()
    _ops.unregister_hook_predicate(a)
    assert not _ops.grad_hooks_active()
    _ops.unregister_hook_predicate(b)
    assert not _ops._hook_predicates


def test_packed_bucket_plan_host_logic():
    """dp.DataParallelStep._packed_key (no GPU needed): per modality the packed row count is rounded up to the next multiple of the
    bucket size and always leaves at least one surplus row, never more than one full sequence of them; cu_seqlens = cumulative
    valid lengths + the bucket's row count; lengths handed over and lengths read from the masks give the same plan; masks that are
    not suffix-PAD, or an empty sequence, are refused"""
    import types
    import pytest
    import torch
    from hri_emo_amd import dp
    B, La, Lt = 8, 400, 128
    stub = types.SimpleNamespace(_pb={"B": B, "La": La, "Lt": Lt, "cu_a": torch.zeros(B + 2, dtype=torch.int32),
                                      "cu_t": torch.zeros(B + 2, dtype=torch.int32)})
    key = lambda m_a, m_t, lens=None: dp.DataParallelStep._packed_key(stub, m_a, m_t, lens)
    g = torch.Generator().manual_seed(1)
    ga, gt = max(8, (B * La) // dp._VARLEN_BUCKETS // 8 * 8), max(8, (B * Lt) // dp._VARLEN_BUCKETS // 8 * 8)
    for trial in range(20):
        la = torch.randint(1, La + 1, (B,), generator=g)
        lt = torch.randint(1, Lt + 1, (B,), generator=g)
        if trial == 0:
            la[:], lt[:] = La, Lt                       # everything full length
        if trial == 1:
            la[:], lt[:] = ga // 2, gt // 2             # valid rows an exact multiple of the bucket size
        m_a, m_t = torch.arange(La)[None] >= la[:, None], torch.arange(Lt)[None] >= lt[:, None]
        ra, rt = key(m_a, m_t)
        cu_a, cu_t = stub._pb["cu_a"].tolist(), stub._pb["cu_t"].tolist()
        for rows, lens, L, gsz, cu in ((ra, la, La, ga, cu_a), (rt, lt, Lt, gsz_t := gt, cu_t)):
            n = int(lens.sum())
            assert rows % min(gsz, L) == 0 and n < rows <= n + min(gsz, L) and rows - n <= L, (trial, rows, n)
            assert cu[0] == 0 and cu[1:B + 1] == torch.cumsum(lens, 0).tolist() and cu[B + 1] == rows
        assert key(m_a, m_t, (la.tolist(), lt.tolist())) == (ra, rt)
        assert stub._pb["cu_a"].tolist() == cu_a and stub._pb["cu_t"].tolist() == cu_t
    hole = m_a.clone(); hole[0, 0] = True; hole[0, 1] = False
    with pytest.raises(RuntimeError, match="suffix"):
        key(hole, m_t)
    empty = m_t.clone(); empty[3, :] = True
    with pytest.raises(RuntimeError, match="suffix"):
        key(m_a, empty)
    with pytest.raises(ValueError, match="lengths"):
        key(m_a, m_t, ([0] * B, [1] * B))


def test_bench_becomes_its_own_launcher_for_several_gpus(monkeypatch):
    """`python bench.py --gpus N` without WORLD_SIZE: bench.py starts torch.distributed.run as a child process (one rank per GPU,
    rendezvous on 127.0.0.1) with its own arguments and returns the child's exit code -- before anything touches the GPU."""
    import importlib.util
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(repo, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    class R:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return R()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert os.path.basename(cmd[-7]) == "bench.py" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_step_context_is_per_step_and_restored():
    """_ops.StepContext: the capture flag / capture stream / packed plan / join scope / half-gradient bookkeeping of a step in
    flight live in the context its DataParallelStep owns, installed by use_context and restored on the way out (re-entrant), so
    two steps in one process do not see each other's state."""
    import hri_emo_amd  # noqa: F401
    from hri_emo_amd import _ops
    base = _ops.CTX
    a, b = _ops.StepContext(), _ops.StepContext()
    assert a.half_reports is not b.half_reports and not a.capturing and a.seq_override is None and a.join_scope == 0
    with _ops.use_context(a):
        assert _ops.CTX is a
        _ops.CTX.capturing = True
        _ops.CTX.join_scope += 1
        with _ops.use_context(b):
            assert _ops.CTX is b and not _ops.CTX.capturing and _ops.CTX.join_scope == 0
            _ops.CTX.half_reports[1] = 2
        assert _ops.CTX is a and a.capturing and a.join_scope == 1 and not a.half_reports and b.half_reports == {1: 2}
        try:
            with _ops.use_context(b):
                raise KeyError("x")
        except KeyError:
            pass
        assert _ops.CTX is a                       # restored on an exception too
    assert _ops.CTX is base and not base.capturing
    from hri_emo_amd import dp
    assert dp.DataParallelStep.step.__wrapped__ is not None and dp.DataParallelStep.capture.__wrapped__ is not None
