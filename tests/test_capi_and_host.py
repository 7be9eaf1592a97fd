"""CPU-only checks: the C-ABI library loads and exports every symbol declared in include/mau_hip.h
(no compute calls without a GPU), the Python binding mirrors the header, and the host-side module
keeps the reference's constructor / state_dict / error contract."""
import os
import re

import numpy as np

import pytest
import torch

from oracle import unet_ref as R
from tests.helpers import load_npz, meta_of, sub

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "mau_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mau_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_header_symbol():
    import ctypes
    import mau_amd
    from mau_amd import _lib
    syms = header_symbols()
    assert len(syms) >= 35
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), f"libmau_hip.so does not export {s}"
    assert sorted(_lib.PROTOTYPES) == syms, (set(_lib.PROTOTYPES) ^ set(syms))
    assert _lib.lib.mau_abi_version() == 5
    # pure host-side helpers of the ABI are callable without a GPU
    assert _lib.lib.mau_conv3x3_kc(_lib.MAU_BF16) == 16 and _lib.lib.mau_conv3x3_kc(_lib.MAU_F32) == 16
    assert _lib.lib.mau_conv3x3_packed_elems(_lib.MAU_BF16, 64, 6) == 1 * 9 * 64 * 16
    # bf16 tile height is chosen per layer (grid fill): 64-row (64-wide) / 32-row (128-wide) tiles on a full grid, 16-row
    # tiles when the layer is small (slab rows = workgroup tiles x wave rows per workgroup)
    assert _lib.lib.mau_conv3x3_num_pixel_tiles(_lib.MAU_BF16, 32, 256, 256, 64) == 32 * 4 * 16 * 8
    assert _lib.lib.mau_conv3x3_num_pixel_tiles(_lib.MAU_BF16, 32, 32, 32, 256) == 32 * 2 * 2 * 4
    # a 128-multiple layer that would leave most CUs idle runs on 64-channel workgroups: 8-row tiles while even their items are fewer
    # than half of the CUs (2 images x 2 x 1 tiles x 4 wave rows), 16-row tiles otherwise (16 images x 1 x 1 x 4)
    assert _lib.lib.mau_conv3x3_num_pixel_tiles(_lib.MAU_BF16, 2, 16, 16, 1024) == 2 * 2 * 4
    assert _lib.lib.mau_conv3x3_num_pixel_tiles(_lib.MAU_BF16, 16, 16, 16, 1024) == 16 * 4
    assert _lib.lib.mau_conv3x3_num_pixel_tiles(_lib.MAU_BF16, 32, 16, 16, 1024) == 32 * 4
    assert _lib.lib.mau_conv3x3_num_pixel_tiles(_lib.MAU_BF16, 32, 16, 16, 1024) == 32 * 4
    assert _lib.lib.mau_conv3x3_num_pixel_tiles(_lib.MAU_F32, 2, 250, 250, 64) == 2 * 32 * 16
    # split-K count of the weight gradient: the fitted cost model rounds(s) * (nTiles / s + overhead) -- one round of 256 workgroups where
    # the layer has few (co,ci) tiles, FEWER splits than whole rounds would take on the K-heavy layers (profiles/r5/wgrad_split_probe.txt)
    assert _lib.lib.mau_conv3x3_wgrad_splits(_lib.MAU_BF16, 32, 256, 256, 64, 64) == 256
    # (round 6: every split count has an XCD-contiguous work-item order, so counts that are not whole numbers per XCD compete on equal terms)
    assert _lib.lib.mau_conv3x3_wgrad_splits(_lib.MAU_BF16, 32, 32, 32, 512, 1536) == 5      # 96 (co,ci) tiles: 480 workgroups in two rounds instead of 768 in three
    assert _lib.lib.mau_conv3x3_wgrad_splits(_lib.MAU_BF16, 32, 64, 64, 256, 768) == 10      # 24 tiles: 240 workgroups instead of 192
    assert _lib.lib.mau_conv3x3_wgrad_splits(_lib.MAU_BF16, 32, 16, 16, 1024, 576) == 3
    # the tile variant a layer runs, and the number of K groups per workgroup (2 = the under-filled-layer form: at most one workgroup
    # per CU, an even stage count >= 4; needs the layer's input channels)
    assert _lib.conv3x3_variant(_lib.MAU_BF16, 32, 256, 256, 64) == (32, 4, 64, 1) and _lib.conv3x3_variant(_lib.MAU_BF16, 32, 128, 128, 128) == (32, 8, 128, 1)
    assert _lib.conv3x3_variant(_lib.MAU_BF16, 1, 32, 32, 1024) == (8, 4, 64, 1) and _lib.conv3x3_variant(_lib.MAU_F32, 2, 31, 17, 70)[2] == 64
    assert _lib.conv3x3_variant(_lib.MAU_BF16, 1, 32, 32, 1024, Cin=1024) == (8, 4, 64, 2)      # B = 1 conv4_0.conv2: 128 items on 256 CUs
    assert _lib.conv3x3_variant(_lib.MAU_BF16, 8, 32, 32, 1024, Cin=1024)[3] == 1              # B = 8: the chip is full, one K group
    assert _lib.conv3x3_variant(_lib.MAU_BF16, 1, 32, 32, 1024, Cin=48)[3] == 1                # three stages: odd, no K groups
    # the fp32 parity mode also writes split-K partial slabs (plain stores + fixed-order sum: no float atomics anywhere)
    s32 = _lib.lib.mau_conv3x3_wgrad_splits(_lib.MAU_F32, 32, 32, 32, 512, 1536)
    assert 1 <= s32 <= 32 * 4 * 2 and s32 * 9 * 512 * 1536 * 4 <= 256 << 20
    assert _lib.lib.mau_conv3x3_wgrad_acc_elems(_lib.MAU_F32, 32, 32, 32, 512, 1536) == s32 * 9 * 512 * 1536
    assert _lib.lib.mau_reduce_tickets_elems() >= 2 * 1024 // 64


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    """The product path has no fallback: without libmau_hip.so the import itself raises."""
    import importlib.util
    import mau_amd
    src = os.path.join(os.path.dirname(mau_amd._lib.__file__), "_lib.py")
    dst = tmp_path / "_lib_copy.py"
    dst.write_text(open(src).read())
    spec = importlib.util.spec_from_file_location("_lib_copy", dst)
    mod = importlib.util.module_from_spec(spec)
    with pytest.raises(ImportError, match="no CPU or PyTorch fallback"):
        spec.loader.exec_module(mod)


def test_cpu_tensors_are_refused():
    import mau_amd
    net = mau_amd.UrbanPredictor("unet", 6, 10, 8, 4, 8, 12, 2, base_filters=4, temporal_embeddings=False)
    with pytest.raises(RuntimeError, match="no CPU"):
        net(torch.zeros(1, 6, 16, 16), torch.zeros(1, 10), torch.zeros(1, 4))


def test_constructor_and_state_dict_contract():
    import mau_amd
    # dispatcher errors (src/model.py:326) and kwargs behaviour (:196-200 vs :52-53)
    with pytest.raises(ValueError, match="Unsupported model_type"):
        mau_amd.UrbanPredictor("resnet", 6, 10, 64, 4, 64, 96, 2)
    with pytest.raises(TypeError):
        mau_amd.UrbanPredictor("unet", 6, 10, 64, 4, 64, 96, 2, not_a_flag=True)
    mau_amd.UrbanPredictor("unet++", 6, 10, 8, 4, 8, 12, 2, base_filters=4, temporal_embeddings=False)   # swallowed by **kwargs
    # 138 / 222 state entries and the parameter counts of SURVEY 8(b)
    torch.manual_seed(0)
    net = mau_amd.UrbanPredictor("unet", 6, 10, 64, 4, 64, 96, 2, temporal_embeddings=False, metadata_embeddings=True)
    assert len(net.state_dict()) == 138 and sum(p.numel() for p in net.parameters()) == 32028834
    assert net.state_dict()["model.conv0_0.bn1.num_batches_tracked"].dtype == torch.int64
    netpp = mau_amd.UrbanPredictor("unet++", 6, 10, 64, 4, 64, 96, 2)
    assert len(netpp.state_dict()) == 222 and sum(p.numel() for p in netpp.parameters()) == 38594850


@pytest.mark.parametrize("mt,flags", [("unet", dict(temporal_embeddings=False, metadata_embeddings=True)),
                                      ("unet", dict(temporal_embeddings=True, metadata_embeddings=False)), ("unet++", {})])
def test_same_seed_same_initial_weights_as_reference(mt, flags):
    import mau_amd
    torch.manual_seed(7)
    net = mau_amd.UrbanPredictor(mt, 6, 10, 8, 4, 8, 12, 2, base_filters=4, **flags)
    torch.manual_seed(7)
    ref = R.init_state(mt, 6, 10, 8, 4, 8, 12, 2, base_filters=4, **flags)
    sd = net.state_dict()
    assert set(sd) == set(ref)
    for k in ref:
        assert torch.equal(sd[k], ref[k]), k


@pytest.mark.parametrize("name", ["g5_unet_even.npz", "g6_unetpp.npz"])
def test_reference_checkpoints_load_strict(name):
    """state_dicts produced by the REFERENCE (fixtures) load with strict=True, and round-trip through
    torch.save/torch.load in the reference's checkpoint dict layout (src/train.py:305-316)."""
    import io
    import mau_amd
    d = load_npz(name)
    kw = meta_of(d)["kw"]
    net = mau_amd.UrbanPredictor(**kw)
    ref_sd = sub(d, "sd0")
    net.load_state_dict(ref_sd, strict=True)
    bad = dict(ref_sd)
    bad.pop("model.final.bias")
    with pytest.raises(RuntimeError):
        net.load_state_dict(bad, strict=True)
    ckpt = {"epoch": 0, "step": 1, "model_state_dict": net.state_dict(), "optimizer_state_dict": {}, "loss": 0.0,
            "hyperparameters": {}, "model_type": kw["model_type"], "study_name": "s", "trial_id": 0, "metadata_input_length": kw["meta_features"]}
    buf = io.BytesIO()
    torch.save(ckpt, buf)
    buf.seek(0)
    back = torch.load(buf, weights_only=False)
    for k, v in ref_sd.items():
        assert torch.equal(back["model_state_dict"][k], v), k


def test_precision_switch_and_env(monkeypatch):
    import mau_amd
    net = mau_amd.UrbanPredictor("unet", 6, 10, 8, 4, 8, 12, 2, base_filters=4)
    assert net.model._rt.precision in ("bf16", "fp32")
    net.set_precision("fp32")
    assert net.model._rt.dtype == torch.float32 and net.model.conv0_0._rt is net.model._rt
    with pytest.raises(ValueError):
        net.set_precision("fp8")
    monkeypatch.setenv("MAU_PRECISION", "fp32")
    assert mau_amd.UrbanPredictor("unet", 6, 10, 8, 4, 8, 12, 2, base_filters=4).model._rt.precision == "fp32"


def test_checkpoint_reader_rules_and_roundtrip(tmp_path):
    """Reader resolution rules of app/model_utils.py:38-74 / test/evaluate.py:85-113 and the writer's dict
    (src/train.py:305-316); construction + strict load work on CPU (only forward needs the GPU)."""
    import mau_amd
    from mau_amd import checkpoint as C
    from mau_amd.config import CONFIG
    f = C.resolve_embedding_flags
    assert f({"hyperparameters": {"temporal_embeddings": False, "metadata_embeddings": True}}) == (False, True)
    assert f({}) == (True, True)                                                   # legacy default
    assert f({"study_name": "urban-predictor-noemb"}) == (False, False)
    assert f({"additional_embeddings": False, "metadata_only_embeddings": True}) == (False, True)
    assert f({}, study_name="x-noemb") == (False, False)
    kw = C.model_kwargs_from_checkpoint({})
    assert (kw["model_type"], kw["meta_features"], kw["temporal_dim"], kw["meta_dim"], kw["lstm_dim"]) == ("unet", 4, 64, 64, 96)
    assert CONFIG.training.learning_rate == 1e-4 and CONFIG.dataset.nb_input_channels == 23
    # write with the reference's layout, read back through the loader mirror
    net = mau_amd.UrbanPredictor("unet", 23, 10, 8, 8, 8, 12, 2, base_filters=64, temporal_embeddings=False, metadata_embeddings=True)
    hyper = {"temporal_dim": 8, "meta_dim": 8, "lstm_hidden": 12, "temporal_embeddings": False, "metadata_embeddings": True}
    path = str(tmp_path / "m.pth")
    ck = C.save_checkpoint(path, net, None, epoch=1, step=2, loss=0.5, hyperparameters=hyper, model_type="unet",
                           study_name="urban-predictor-metaemb", trial_id=3, metadata_input_length=8)
    assert list(ck) == ["epoch", "step", "model_state_dict", "optimizer_state_dict", "loss", "hyperparameters", "model_type",
                        "study_name", "trial_id", "metadata_input_length"]
    back = C.load_model(path, device="cpu", clean=True)
    assert not back.training and back.model.temporal_embeddings is False and back.model.metadata_embeddings is True
    for k, v in net.state_dict().items():
        assert torch.equal(back.state_dict()[k], v), k
    assert "optimizer_state_dict" not in torch.load(path, weights_only=False)       # the reference's clean-up side effect
    # bare state_dict and 'state_dict' variants (app/model_utils.py:91-96)
    torch.save({"state_dict": net.state_dict(), "hyperparameters": hyper, "metadata_input_length": 8}, path)
    C.load_model(path, device="cpu")


# --------------------------------------------------------------------------- #
# input pipeline (host logic; the device kernel is covered in tests/test_gpu_ops.py)
# --------------------------------------------------------------------------- #
def _fake_tile(rng, H=20, W=18, nc=9):
    a = rng.integers(0, nc, (H, W)); b = rng.integers(0, nc, (H, W))
    cont = rng.standard_normal((5, H, W)).astype(np.float32)
    dense = np.vstack([np.eye(nc)[a].transpose(2, 0, 1), cont, np.eye(nc)[b].transpose(2, 0, 1)]).astype(np.float32)   # process.py:181
    return a, b, cont, dense


def test_compact_input_roundtrip_and_validation():
    import mau_amd
    rng = np.random.default_rng(0)
    a, b, cont, dense = _fake_tile(rng)
    ca, cb, cc = mau_amd.data.compact_input(dense)
    assert ca.dtype == np.uint8 and np.array_equal(ca, a) and np.array_equal(cb, b) and np.array_equal(cc, cont)
    assert np.array_equal(mau_amd.data.expand_input(ca, cb, cc), dense)
    bad = dense.copy(); bad[0, 0, 0] = 0.5
    with pytest.raises(ValueError):
        mau_amd.data.compact_input(bad)
    bad = dense.copy(); bad[:9, 3, 3] = 0
    with pytest.raises(ValueError):
        mau_amd.data.compact_input(bad)


def test_random_flip_draws_follow_the_reference_stream():
    """src/dataset.py:134-141: random.seed(seed) at construction, one random.random() < 0.5 per sample."""
    import random
    import mau_amd
    rf = mau_amd.data.RandomFlip(seed=42)
    got = [rf.draw() for _ in range(64)]
    random.seed(42)
    assert got == [random.random() < 0.5 for _ in range(64)] and 10 < sum(got) < 54
    rf = mau_amd.data.RandomFlip(seed=7)
    x = np.arange(2 * 3 * 4, dtype=np.float32).reshape(2, 3, 4); y = x[:1] * 2
    random.seed(7); want = random.random() < 0.5
    fx, fy = rf(x, y)
    assert np.array_equal(fx, np.flip(x, 2) if want else x) and np.array_equal(fy, np.flip(y, 2) if want else y)


def test_dataset_and_collate_mirror_the_reference_contract(tmp_path):
    import mau_amd
    rng = np.random.default_rng(1)
    d = tmp_path / "train"; d.mkdir()
    dense = {}
    for i, n_ts in enumerate((12, 9, 12)):
        a, b, cont, x = _fake_tile(rng)
        name = f"Some City_{i}_48.8566_2.3522_2019_0{i + 1}_to_2021_0{i + 2}.npz"     # process.py:158
        np.savez_compressed(d / name, input=x, target=rng.standard_normal((2, 20, 18)).astype(np.float32),
                            metadata=rng.standard_normal(4).astype(np.float32), temperature_serie=rng.standard_normal(n_ts).astype(np.float32))
        dense[name] = x
    with pytest.raises(FileNotFoundError):
        mau_amd.data.FuturePredictionDataset("val", processed_dir=str(tmp_path))
    ref_like = mau_amd.data.FuturePredictionDataset("train", processed_dir=str(tmp_path), compact=False)
    assert len(ref_like) == 3
    x0, md0, ts0, t1, t2, tg0 = ref_like[0]
    assert x0.shape == (23, 20, 18) and t1.tolist() == [2019.0, 1.0] and t2.tolist() == [2021.0, 2.0] and tg0.shape == (2, 20, 18)
    assert ref_like.get_metadata_from_idx(0) == {"city": "Some City", "lat": 48.8566, "lon": 2.3522}
    ds = mau_amd.data.FuturePredictionDataset("train", processed_dir=str(tmp_path), transform=mau_amd.data.RandomFlip(3))
    batch = mau_amd.data.collate_fn([ds[i] for i in range(3)])
    assert batch.cls_a.dtype == torch.uint8 and batch.cls_a.shape == (3, 20, 18) and batch.cont.shape == (3, 5, 20, 18)
    assert batch.temp_series.shape == (3, 12) and batch.temp_series_lengths.tolist() == [12, 9, 12] and float(batch.temp_series[1, 9:].abs().sum()) == 0
    assert batch.flip.dtype == torch.uint8 and batch.targets.shape == (3, 2, 20, 18)
    for i in range(3):
        x = mau_amd.data.expand_input(batch.cls_a[i].numpy(), batch.cls_b[i].numpy(), batch.cont[i].numpy())
        assert np.array_equal(x, ref_like[i][0].numpy())
    dense_bytes = 3 * (23 + 2) * 20 * 18 * 4
    assert batch.host_bytes() < 0.4 * dense_bytes


def test_asan_host_build_of_the_c_abi():
    """`make asan` (csrc/Makefile): the library's HOST code (argument validation, size helpers, error reporting) built
    with -fsanitize=address, driven by tests/asan_host_check.c -- every size helper and one refused call per entry-point
    family -- without a GPU.  (No GPU-side sanitizer exists on this pool: SURVEY section 5.)"""
    import shutil
    import subprocess
    if not os.path.exists(subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()) \
            or shutil.which("make") is None:
        pytest.skip("no libasan / make in this environment")
    csrc = os.path.join(ROOT, "metadata-augmented-unet-for-lst-ndvi_amd", "csrc")
    p = subprocess.run(["make", "-C", csrc, "asan"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "asan host check OK" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_live_autograd_graph_probe_cpu():
    """train_graph.live_autograd_graph_params (the guard in front of every capture): a parameter's AccumulateGrad node survives
    only while some autograd graph holds it."""
    import mau_amd  # noqa: F401
    from mau_amd.train_graph import live_autograd_graph_params
    lin = torch.nn.Linear(4, 4)
    assert live_autograd_graph_params(lin.parameters()) == []
    kept = lin(torch.randn(2, 4)).sum()
    assert live_autograd_graph_params(lin.parameters()) == [0, 1]
    del kept
    assert live_autograd_graph_params(lin.parameters()) == []
    hooks = [p.register_post_accumulate_grad_hook(lambda p: None) for p in lin.parameters()]      # (dist.GradSync's hooks pin nothing)
    lin(torch.randn(2, 4)).sum().backward()
    assert live_autograd_graph_params(lin.parameters()) == []


def test_pmc_record_belongs_to_the_library_in_the_tree():
    """bench.py quotes `roofline.traffic` / `mfma_busy` from profiles/<round>/pmc_summary.json only when that record carries the
    sha256 of the libmau_hip.so it loaded.  This test says, on the CPU, whether the committed record still belongs to the library
    the sources build (the build is reproducible: same toolchain, same bytes); a kernel change without new records shows up here
    as a SKIP with the reason -- bench.py then reports `traffic: null` and says why -- instead of going unnoticed."""
    import glob
    import hashlib
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "metadata-augmented-unet-for-lst-ndvi_amd", "libmau_hip.so")
    if not os.path.exists(so):
        pytest.skip("libmau_hip.so not built")
    have = hashlib.sha256(open(so, "rb").read()).hexdigest()
    recs = sorted(glob.glob(os.path.join(root, "profiles", "r*", "pmc_summary.json")))
    assert recs, "no PMC record committed under profiles/"
    latest = json.load(open(recs[-1]))
    assert latest and all("lib_sha256" in v for v in latest.values()), "a PMC record without the library's sha256"
    stale = {k: v["lib_sha256"][:12] for k, v in latest.items() if v["lib_sha256"] != have}
    if stale:
        pytest.skip(f"{recs[-1]} was measured on another build ({stale}; this tree builds {have[:12]}): re-run scripts/r4_final.sh + r4_records2.sh")


def test_hand_counted_lds_waits_hold_in_the_emitted_isa():
    """ADVICE r4 (conv3x3_first.hip ``tr_frag``, conv3x3_wgrad16 / _wgrad_bf16 ``tr_read``): fragment reads issued from inline asm are
    invisible to the compiler's wait-count pass; that no copy or consumer of a fragment register sits in front of the hand-counted
    ``s_waitcnt lgkmcnt(N)`` is a property of the register allocation of THIS build.  ``scripts/check_lds_waits.py`` walks the
    disassembly of every kernel of the in-tree objects with the hardware's in-order LDS model and must find no such instruction."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_lds_waits", os.path.join(ROOT, "scripts", "check_lds_waits.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    # the checker itself: a copy of a transposed-read half in front of the wait IS reported, behind it is not; counted waits are honoured
    bad = "0000000000001000 <k>:\n\tds_read_b64_tr_b16 v[10:11], v1  // 000000001000: 0\n\tds_read_b64_tr_b16 v[12:13], v1 offset:512  // 000000001008: 0\n" \
          "\tv_mov_b32_e32 v20, v12  // 000000001010: 0\n\ts_waitcnt lgkmcnt(0)  // 000000001014: 0\n\tv_mov_b32_e32 v21, v10  // 000000001018: 0\n\ts_endpgm  // 00000000101c: 0\n"
    h, r = chk.check(bad, "synthetic")
    assert r == 2 and len(h) == 1 and "v20, v12" in h[0][2]
    counted = "0000000000001000 <k>:\n\tds_read_b128 v[0:3], v9  // 000000001000: 0\n\tds_read_b128 v[4:7], v9 offset:16  // 000000001008: 0\n" \
              "\ts_waitcnt lgkmcnt(1)  // 000000001010: 0\n\tv_mfma_f32_16x16x32_bf16 v[20:23], v[0:3], v[0:3], v[20:23]  // 000000001014: 0\n" \
              "\tv_mfma_f32_16x16x32_bf16 v[20:23], v[4:7], v[4:7], v[20:23]  // 00000000101c: 0\n\ts_endpgm  // 000000001024: 0\n"
    h, r = chk.check(counted, "synthetic")
    assert r == 2 and len(h) == 1 and "v[4:7]" in h[0][2]            # the first fragment has landed, the second has not
    objs = [o for o in __import__("glob").glob(os.path.join(ROOT, "metadata-augmented-unet-for-lst-ndvi_amd", "csrc", "*.o")) if not o.endswith(".asan.o")]
    assert len(objs) >= 15, "build the library first (python -c 'import __graft_entry__ as g; g.build()')"
    import tempfile
    total, hazards = 0, []
    with tempfile.TemporaryDirectory() as tmp:
        for o in sorted(objs):
            asm = chk.disassemble(o, tmp)
            if asm:
                h, r = chk.check(asm, os.path.basename(o))
                total += r
                hazards += h
    assert total > 10000, total                                        # (the convolution kernels alone issue thousands of fragment reads)
    assert not hazards, hazards[:5]
    # ... and no matrix-core kernel touches scratch INSIDE its multiply loops (between two MFMAs): a spilled register comes back through
    # a vector-memory load whose latency would stall the matrix pipes and whose vmcnt slot the hand-counted DMA waits do not know
    # about.  (Round 5 found 8-26 spilled registers in several instantiations of the convolution kernel -- loop-invariant slot
    # coordinates of the loader, hoisted out of the item loop: reloaded at the head and the tail of an item, outside the multiply
    # loop; the inference-epilogue forms now recompute them and spill nothing.)
    kernels, in_loop, spilling = 0, [], []
    hot = ("conv3x3_bf16_kernel", "wgrad16_kernel", "wgrad_bf16_kernel", "first_fwd_kernel", "first_wgrad_kernel", "conv3x3_igemm_kernel", "conv3x3_wgrad_kernel")
    with tempfile.TemporaryDirectory() as tmp:
        for o in sorted(objs):
            res = chk.kernel_resources(o, tmp)
            kernels += len(res)
            spilling += [(n, v, sp, sc) for n, v, sp, sc in res if any(t in n for t in hot) and sc]
            if any(sc for n, v, sp, sc in res if any(t in n for t in hot)):
                in_loop += [x for x in chk.scratch_in_mfma_region(chk.disassemble(o, tmp)) if any(t in x[0] for t in hot)]
    assert kernels > 300, kernels
    assert not in_loop, in_loop[:5]
    assert not [k for k in spilling if "Li2ELb" in k[0] and "conv3x3_bf16_kernel" in k[0] and "ELb1ELb1E" in k[0]], spilling      # (inference-epilogue big-tile forms: none)
    assert len(spilling) <= 16 and max([k[2] for k in spilling] or [0]) <= 16, spilling
    bad = "0000000000001000 <k>:\n\tv_mfma_f32_16x16x32_bf16 v[0:3], v[4:7], v[8:11], v[0:3]  // 000000001000: 0\n\tscratch_load_dword v9, off, off  // 000000001008: 0\n" \
          "\tv_mfma_f32_16x16x32_bf16 v[0:3], v[4:7], v[8:11], v[0:3]  // 000000001010: 0\n\tscratch_load_dword v9, off, off  // 000000001018: 0\n\ts_endpgm  // 00000000101c: 0\n"
    assert len(chk.scratch_in_mfma_region(bad)) == 1                   # (the access behind the last MFMA is outside the multiply region)
