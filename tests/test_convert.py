"""Ultralytics state dict -> VTIW1 container (BN folding + naming).

No Ultralytics checkpoint exists here (the reference's .pt files are absent blobs and the package is not
installable), so the state dict is synthetic but uses Ultralytics' tensor names;
folding is checked against torch's own BatchNorm2d in eval mode, layer by layer, and the converted
container must drive the CPU oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F


def fake_state_dict(eng, seed=0):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for t in eng.conv_table():
        n, c1, c2, k, kind = t["name"], t["c1"], t["c2"], t["k"], t["kind"]
        if kind == 0:
            sd[n + ".conv.weight"] = torch.randn((c2, c1, k, k), generator=g) / (c1 * k * k) ** 0.5
            sd[n + ".bn.weight"] = torch.rand(c2, generator=g) + 0.5
            sd[n + ".bn.bias"] = torch.randn(c2, generator=g) * 0.1
            sd[n + ".bn.running_mean"] = torch.randn(c2, generator=g) * 0.2
            sd[n + ".bn.running_var"] = torch.rand(c2, generator=g) + 0.25
            sd[n + ".bn.num_batches_tracked"] = torch.tensor(1)
        else:
            shape = (c1, c2, k, k) if kind == 2 else (c2, c1, k, k)
            sd[n + ".weight"] = torch.randn(shape, generator=g) / (c1 * k * k) ** 0.5
            sd[n + ".bias"] = torch.randn(c2, generator=g) * 0.1
    sd["model.22.dfl.conv.weight"] = torch.arange(16, dtype=torch.float32).view(1, 16, 1, 1)
    return sd


def test_bn_folding_matches_torch_batchnorm(lib_built):
    vti = lib_built
    eng = vti.Engine("n", 2, H=64, W=64, max_batch=1)              # the reference's model: 2 classes (config.py:69-70)
    sd = fake_state_dict(eng)
    blob = vti.convert_state_dict(sd, eng)
    meta, table, tensors = vti.unpack_container(blob)
    assert meta == dict(scale="n", nc=2, nm=32, reg_max=16) and [t["name"] for t in table] == [t["name"] for t in eng.conv_table()]
    g = torch.Generator().manual_seed(5)
    for t in eng.conv_table():
        n, c1, c2, k, s, kind = t["name"], t["c1"], t["c2"], t["k"], t["s"], t["kind"]
        w, b = (torch.from_numpy(np.array(a)) for a in tensors[n])
        x = torch.randn((1, c1, 9, 11), generator=g)
        if kind == 0:
            bn = torch.nn.BatchNorm2d(c2, eps=1e-3).eval()
            bn.weight.data, bn.bias.data = sd[n + ".bn.weight"], sd[n + ".bn.bias"]
            bn.running_mean, bn.running_var = sd[n + ".bn.running_mean"], sd[n + ".bn.running_var"]
            ref = bn(F.conv2d(x, sd[n + ".conv.weight"], None, s, k // 2))
            got = F.conv2d(x, w, b, s, k // 2)
            assert torch.allclose(got, ref, rtol=1e-4, atol=2e-5), n      # fp32 summation order only
        elif kind == 1:
            assert torch.equal(w, sd[n + ".weight"]) and torch.equal(b, sd[n + ".bias"]), n
        else:
            assert tuple(w.shape) == (c1, c2, 2, 2) and torch.equal(w, sd[n + ".weight"]), n


def test_converted_container_drives_the_oracle_and_rejects_a_wrong_plan(lib_built):
    from oracle.model import OracleModel
    vti = lib_built
    eng = vti.Engine("n", 2, H=64, W=64, max_batch=1)
    blob = vti.convert_state_dict(fake_state_dict(eng), eng)
    frames = np.random.Generator(np.random.PCG64(0)).integers(0, 256, (1, 64, 64, 3), dtype=np.uint8)
    pred, proto = OracleModel(blob, 64, 64, "fp32").forward_u8(frames, swap_rb=True)
    assert tuple(pred.shape) == (1, 4 + 2 + 32, 8 * 8 + 4 * 4 + 2 * 2) and tuple(proto.shape) == (1, 32, 16, 16)
    assert torch.isfinite(pred).all() and torch.isfinite(proto).all()
    other = vti.Engine("n", 80, H=64, W=64, max_batch=1)            # checkpoint has 2 classes, plan has 80
    with pytest.raises(ValueError, match="model.22.cv3"):
        vti.convert_state_dict(fake_state_dict(eng), other)
    sd = fake_state_dict(eng)
    del sd["model.4.m.1.cv2.bn.running_var"]
    with pytest.raises(KeyError, match="model.4.m.1.cv2.bn.running_var"):
        vti.convert_state_dict(sd, eng)
