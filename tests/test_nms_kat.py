"""Hand-derived NMS known answers (tests/nms_kat.py) against the oracle (CPU) and the HIP kernel (-m gpu)."""
import numpy as np
import pytest
import torch

from oracle import yolo11_oracle as oracle

from nms_kat import cases

CASES = cases()


def _check(name, got, exp):
    exp = np.asarray(exp, np.float32).reshape(-1, 6)
    assert got.shape == exp.shape, f"{name}: {got.tolist()} vs {exp.tolist()}"
    np.testing.assert_array_equal(got, exp, err_msg=name)


@pytest.mark.parametrize("name,pred,kw,exp", CASES, ids=[c[0] for c in CASES])
def test_oracle_matches_hand_derived(name, pred, kw, exp):
    got = oracle.non_max_suppression(torch.from_numpy(pred), **kw)[0].numpy()
    _check(name, got, exp)


@pytest.mark.gpu
@pytest.mark.parametrize("name,pred,kw,exp", CASES, ids=[c[0] for c in CASES])
def test_hip_nms_matches_hand_derived(gpu_device, name, pred, kw, exp):
    from yololite_amd.utils import ops as hops

    got = hops.non_max_suppression(torch.from_numpy(pred).to(gpu_device), **kw)[0].cpu().numpy()
    _check(name, got, exp)
