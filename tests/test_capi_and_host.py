"""CPU-only checks: the C-ABI library loads and exports every symbol declared in include/mau_hip.h
(no compute calls without a GPU), the Python binding mirrors the header, and the host-side module
keeps the reference's constructor / state_dict / error contract."""
import os
import re

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
    assert _lib.lib.mau_abi_version() == 1
    # pure host-side helpers of the ABI are callable without a GPU
    assert _lib.lib.mau_conv3x3_kc(_lib.MAU_BF16) == 16 and _lib.lib.mau_conv3x3_kc(_lib.MAU_F32) == 16
    assert _lib.lib.mau_conv3x3_packed_elems(_lib.MAU_BF16, 64, 6) == 1 * 9 * 64 * 16
    assert _lib.lib.mau_conv3x3_num_pixel_tiles(_lib.MAU_BF16, 32, 256, 256, 64) == 32 * 16 * 16
    assert _lib.lib.mau_conv3x3_num_pixel_tiles(_lib.MAU_BF16, 32, 16, 16, 1024) == 32
    assert _lib.lib.mau_conv3x3_num_pixel_tiles(_lib.MAU_F32, 2, 250, 250, 64) == 2 * 32 * 16
    s = _lib.lib.mau_conv3x3_wgrad_splits(_lib.MAU_BF16, 32, 32, 32, 512, 1536)
    assert s >= 1 and (4 * 24 * s) % 256 == 0          # whole rounds of 256 CUs
    assert _lib.lib.mau_conv3x3_wgrad_splits(_lib.MAU_F32, 32, 32, 32, 512, 1536) == 1


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
