"""Device-side batch assembly: host row plan on CPU, kernel on the GPU, both against oracle/batch_oracle.py."""
import random

import numpy as np
import pytest

from oracle import batch_oracle as BO

SPECS = [dict(compress_factor=100.0, channels=2), dict(compress_factor=1.0, channels=1), dict(compress_factor=1.0, channels=1)]


def _samples(rng, nbase, missing):
    out = []
    for i in range(nbase):
        of = rng.integers(-3000, 3000, (60, 60, 50)).astype(np.int16)
        gray = rng.integers(0, 256, (60, 60, 25)).astype(np.uint8)
        depth = rng.integers(0, 256, (60, 60, 25)).astype(np.uint8)
        row = [of, gray, depth]
        for j in range(3):
            if (i, j) in missing:
                row[j] = None
        out.append(row)
    return out


@pytest.mark.parametrize("expand", [1, 2, 3, 4])
def test_plan_rows_follows_the_generator(expand):
    from ugaitnet_amd.batching import plan_rows
    rng = np.random.default_rng(expand)
    missing = {(1, 0), (2, 2), (4, 1)}
    samples = _samples(rng, 6, missing) if expand == 2 else [[np.zeros((60, 60, 50), np.int16), np.zeros((60, 60, 25), np.uint8),
                                                                np.zeros((60, 60, 25), np.uint8)] for _ in range(6)]
    present = np.array([[s is not None for s in row] for row in samples])
    _, plan_ref = BO.gen_batch_mm(samples, SPECS, expand, seed=77)
    plan = plan_rows(present, expand, rng=random.Random(77))
    assert np.array_equal(plan, plan_ref)


@pytest.mark.parametrize("expand", [1, 2, 3])
def test_plan_rows_two_modalities_follows_the_generator(expand):
    from ugaitnet_amd.batching import plan_rows_2mod
    rng = np.random.default_rng(20 + expand)
    samples = [[s[0], s[1]] for s in _samples(rng, 5, {(1, 0), (3, 1)})]
    present = np.array([[s is not None for s in row] for row in samples])
    _, plan_ref = BO.gen_batch_2mod(samples, SPECS[:2], expand, seed=31)
    assert np.array_equal(plan_rows_2mod(present, expand, rng=random.Random(31)), plan_ref)
    with pytest.raises(ValueError):
        plan_rows_2mod(present, 4)


@pytest.mark.gpu
@pytest.mark.parametrize("expand", [2, 3])
def test_device_assembly_two_modalities(expand):
    from ugaitnet_amd.batching import DeviceBatchAssembler, ModalitySpec, plan_rows_2mod
    rng = np.random.default_rng(40 + expand)
    samples = [[s[0], s[1]] for s in _samples(rng, 4, {(2, 0), (3, 1)})]
    present = np.array([[s is not None for s in row] for row in samples])
    x_ref, _ = BO.gen_batch_2mod(samples, SPECS[:2], expand, seed=6)
    plan = plan_rows_2mod(present, expand, rng=random.Random(6))
    raws = [np.stack([s[j] if s[j] is not None else np.zeros(shape, dt) for s in samples])
            for j, shape, dt in ((0, (60, 60, 50), np.int16), (1, (60, 60, 25), np.uint8))]
    got = DeviceBatchAssembler([ModalitySpec("of", 2, compress_factor=100.0), ModalitySpec("gray", 1)]).assemble(raws, plan, present=present)
    for m in range(2):
        assert np.array_equal(got[m][0].cpu().numpy(), x_ref[2 * m]) and np.array_equal(got[m][1].cpu().numpy(), x_ref[2 * m + 1])


@pytest.mark.gpu
@pytest.mark.parametrize("expand,clip", [(1, (0, 0)), (2, (0, 0)), (3, (2300, 50))])
def test_device_assembly_matches_oracle_bit_for_bit(expand, clip):
    import torch
    from ugaitnet_amd.batching import DeviceBatchAssembler, ModalitySpec, plan_rows
    rng = np.random.default_rng(10 + expand)
    nbase = 5
    missing = {(1, 0), (3, 2)}
    samples = _samples(rng, nbase, missing)
    present = np.array([[s is not None for s in row] for row in samples])
    x_ref, _ = BO.gen_batch_mm(samples, SPECS, expand, seed=5, clip=clip)
    plan = plan_rows(present, expand, rng=random.Random(5))
    raws = []
    for j, shape, dt in ((0, (60, 60, 50), np.int16), (1, (60, 60, 25), np.uint8), (2, (60, 60, 25), np.uint8)):
        raws.append(np.stack([s[j] if s[j] is not None else np.zeros(shape, dt) for s in samples]))
    asm = DeviceBatchAssembler([ModalitySpec("of", 2, compress_factor=100.0), ModalitySpec("gray", 1), ModalitySpec("depth", 1)])
    got = asm.assemble(raws, plan, present=present, clip=clip)
    for m in range(3):
        x, u = got[m]
        assert np.array_equal(x.cpu().numpy(), x_ref[2 * m]), "modality %d payload" % m
        assert np.array_equal(u.cpu().numpy(), x_ref[2 * m + 1]), "modality %d flags" % m
    # the assembled tensors are what GaitCore.forward takes
    assert got[0][0].shape == (nbase * expand, 25, 60, 60, 2) and got[1][0].dtype == torch.float32


@pytest.mark.gpu
def test_device_assembly_rejects_bad_input():
    from ugaitnet_amd.batching import DeviceBatchAssembler, ModalitySpec
    asm = DeviceBatchAssembler([ModalitySpec("gray", 1)])
    with pytest.raises(ValueError):
        asm.assemble([np.zeros((2, 60, 60, 50), np.uint8)], np.zeros((2, 1), np.int32))
    with pytest.raises(ValueError):
        asm.assemble([np.zeros((2, 60, 60, 25), np.uint8)], np.full((2, 1), 5, np.int32))
