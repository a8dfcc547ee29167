"""The label-cycling sampler (ugaitnet_amd/sampler.py) against hand-walked cases of the reference's `__getitem__` loop
(data/mj_dataGeneratorMMUWYHsingle_repetitions.py:149-183)."""
import numpy as np

from ugaitnet_amd.sampler import LabelCyclingSampler


def _records(labels, gaits, per_cell):
    """per_cell records for every (label, gait) pair, in label-major order; returns (targets, gaits)."""
    t, g = [], []
    for l in labels:
        for ga in gaits:
            t += [l] * per_cell
            g += [ga] * per_cell
    return t, g


def test_p_by_k_batches_and_pointer_walk():
    # 3 labels x 2 gait types x 3 records; record id = 6*label_index + 3*gait_index + k
    t, g = _records([10, 20, 30], [1, 2], 3)
    s = LabelCyclingSampler(t, g, batch_size=8, repetition=2, shuffle=False)
    # label 10: gait 1 rec 0, gait 2 rec 3 (pair 1), gait 1 rec 1, gait 2 rec 4 (pair 2 -> next label); then label 20 likewise
    assert s.next_batch() == [0, 3, 1, 4, 6, 9, 7, 10]
    # next batch starts at label 30; its pointers are fresh, label 10's have advanced to record 2 of each gait type
    assert s.next_batch() == [12, 15, 13, 16, 2, 5, 0, 3]      # pointers wrap after the third record
    assert len(s) == 18 // 8


def test_every_batch_is_p_labels_times_2_repetition():
    t, g = _records(list(range(12)), [0, 1, 2], 4)
    s = LabelCyclingSampler(t, g, batch_size=40, repetition=5, shuffle=True, rng=np.random.default_rng(3))
    for _ in range(5):
        ids = s.next_batch()
        labs = [t[i] for i in ids]
        assert len(ids) == 40
        runs = [labs[i:i + 10] for i in range(0, 40, 10)]
        assert all(len(set(r)) == 1 for r in runs) and len({r[0] for r in runs}) == 4   # 4 ids x 10, the CASIA-B batch
        gaits = [g[i] for i in ids[:6]]
        assert gaits == [0, 1, 2, 0, 1, 2]                                               # gait types round-robin


def test_empty_cells_are_skipped_but_counted():
    # label 5 has no record of gait type 2: the visit produces no row but still counts towards the pair (:163-170)
    t = [5, 5, 7, 7, 7, 7]
    g = [1, 1, 1, 1, 2, 2]
    s = LabelCyclingSampler(t, g, batch_size=4, repetition=1, shuffle=False)
    # label 5: gait 1 -> rec 0, gait 2 -> nothing (pair complete, next label); label 7: rec 2, rec 4 (next label);
    # label 5 again: rec 1, nothing; label 7: rec 3 -> batch full
    assert s.next_batch() == [0, 2, 4, 1]
    # the pair counters restart with every batch (:153-154), the label index and the read pointers carry over: label 5
    # again (its gait-1 pointer has wrapped to record 0), then label 7's second records, then label 5's record 1
    assert s.next_batch() == [0, 3, 5, 1]


def test_epoch_end_resets_and_reshuffles_with_the_given_generator():
    t, g = _records([1, 2, 3, 4], [0, 1], 2)
    a = LabelCyclingSampler(t, g, 4, repetition=1, shuffle=True, rng=np.random.default_rng(9))
    b = LabelCyclingSampler(t, g, 4, repetition=1, shuffle=True, rng=np.random.default_rng(9))
    assert [a.next_batch() for _ in range(3)] == [b.next_batch() for _ in range(3)]
    a.on_epoch_end()
    assert a.nextlab_idx == 0 and all(v == 0 for cells in a.gait2ptr.values() for v in cells.values())
    test = LabelCyclingSampler(t, g, 4, repetition=1, shuffle=True, is_test=True)
    assert list(test.ulabs) == [1, 2, 3, 4]                                              # no shuffling at test time (:268)
