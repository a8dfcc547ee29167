"""bench.py's own multi-rank launch (no external torchrun): `--gpus N` must start N ranks or fail loudly, never report a
one-rank number as an N-GPU one."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _backend(world=2):
    """RCCL with one rank per GPU where the box has a GPU per rank, else the gloo rehearsal on the box's one GPU"""
    return "nccl" if torch.cuda.device_count() >= world else "gloo"


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


@pytest.mark.skipif(torch.cuda.device_count() >= 2, reason="needs a box with fewer than 2 GPUs")
def test_more_ranks_than_gpus_is_an_error_not_a_one_rank_run():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0
    assert "GPU" in r.stderr and "--gpus 2" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_world_size_must_match_gpus():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_spawns_its_own_ranks(dev):
    """`python bench.py --gpus 2` alone (gloo rehearsal: both ranks share the box's GPU) prints ONE line with n_gpus 2."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--dense-only",
                        "--clips-per-gpu", "4", "--no-roofline-pass"], env=_env(UGN_DIST_BACKEND=_backend()), capture_output=True,
                       text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["distributed"]["world_size"] == 2
    assert d["config"]["distributed"]["backend"] == _backend() and d["value"] > 0
    cm = d["config"]["distributed"]["collectives_ms_per_step"]          # every rank's view: max / min over the two ranks
    assert "allreduce_grad_ms" in cm and cm["allreduce_grad_ms"]["max_over_ranks"] >= cm["allreduce_grad_ms"]["min_over_ranks"] > 0


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_strong_scaling_splits_the_c4_batch(dev):
    """--workload c4 --scaling strong with 2 ranks: 20 clips per rank, losses on the gathered 40-clip batch."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--workload", "c4", "--scaling", "strong", "--steps", "1", "--warmup",
                        "1", "--no-cpu-baseline", "--dense-only", "--no-roofline-pass"], env=_env(UGN_DIST_BACKEND=_backend()),
                       capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["scaling"] == "strong" and d["config"]["clips_per_gpu"] == 20 and d["config"]["global_batch"] == 40
    assert d["config"]["dp_mode"] == "global"


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_rccl_path_with_one_rank(dev):
    """The RCCL calls of the data-parallel path (process group with device_id, the flat-buffer all-reduce, the bucketed
    all-reduce on the side stream) executed for real -- one rank, because a test box has one GPU."""
    for overlap in ("0", "1"):
        r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--force-dist", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                            "--dense-only", "--clips-per-gpu", "4", "--no-roofline-pass"], env=_env(UGN_AR_OVERLAP=overlap),
                           capture_output=True, text=True, timeout=800)
        assert r.returncode == 0, r.stderr[-3000:]
        d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
        assert d["config"]["distributed"]["backend"] == "nccl" and d["config"]["distributed"]["world_size"] == 1
        assert d["value"] > 0 and d["loss"] == d["loss"]
        # a SCALE record must be diagnosable: milliseconds per step per collective, payload, library version
        cm = d["config"]["distributed"]["collectives_ms_per_step"]
        assert cm and all(v["max_over_ranks"] >= v["min_over_ranks"] >= 0 for v in cm.values())
        assert d["config"]["distributed"]["gradient_bytes"] > 30e6
        assert ("allreduce_exposed_wait_ms" if overlap == "1" else "allreduce_grad_ms") in cm, cm
        # ... and state the launch geometry and the all-reduce mode in force
        di = d["config"]["distributed"]
        assert di["ar_overlap"] == (overlap == "1") and di["persistent_workgroups_in_force"] == 256 and di["dp_mode"] == "replica"


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("dtype", ["f32x3", "h2", "f32"])
def test_roofline_object_of_the_default_command(dev, tmp_path, dtype):
    """The bench line's roofline: dominant kernel by total duration of the serialised pass, fraction of peak on EXECUTED matrix
    FLOPs (<= 1) against the peak of the instruction that executes them, the algorithmic rate beside it, HBM-bound rows priced in
    bytes; --kernel-table writes every row.  f32x3 is what the bare command runs (fp32 tensors, three-way bf16 split: `dtype` starts
    with f32); h2 is the f16x2 line, f32 the Winograd fp32-MFMA line."""
    table = str(tmp_path / "ktable.csv")
    r = subprocess.run([sys.executable, BENCH, "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--dense-only", "--clips-per-gpu",
                        "8", "--kernel-table", table] + ([] if dtype == "f32x3" else ["--dtype", dtype]),
                       env=_env(), capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    roof = d["roofline"]
    assert d["dtype"].startswith("f16x2" if dtype == "h2" else "f32")
    if dtype in ("h2", "f32x3"):
        # both roofs are priced; `bound` names the one whose floor (executed FLOPs / 2516.8 TFLOP/s, algorithmic bytes / 8 TB/s) is higher
        assert roof["bound"] in ("mfma", "hbm") and 0 < roof["mfma_frac"] <= 1.0 and 0 < roof["hbm_frac"] <= 1.0
        assert (roof["bound"] == "mfma") == (roof["mfma_floor_us"] >= roof["hbm_floor_us"])
        assert roof["frac"] == (roof["mfma_frac"] if roof["bound"] == "mfma" else roof["hbm_frac"])
        assert (roof["unit"], roof["peak"]) == (("TFLOP/s", 2516.8) if roof["bound"] == "mfma" else ("GB/s", 8000.0))
        # three f16 (h2) / six bf16 (x3) MFMAs per product; the pooled weight gradients on the sparse pipe execute half of them
        ratios = (3.0, 1.5) if dtype == "h2" else (6.0, 3.0)
        assert min(abs(roof["mfma_tflops"] / roof["algorithmic_tflops"] - r) for r in ratios) < 0.01
        if dtype == "f32x3":     # the contract's literal figure beside it: algorithmic FLOPs against the fp32-MFMA peak (may exceed 1)
            assert abs(roof["algorithmic_frac_of_f32_mfma_peak"] - roof["algorithmic_tflops"] / 157.3) < 2e-3
            assert abs(d["whole_step_frac_of_f32_mfma_peak"] - d["whole_step_tflops"] / 157.3) < 2e-3
    else:
        assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and roof["peak"] == 157.3
        assert abs(roof["achieved"] / roof["algorithmic_tflops"] - 16.0 / 36.0) < 0.01     # Winograd: 16/36 of the direct count
    assert 0.05 < roof["frac"] <= 1.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    names = {"h2": ("_mm_kernel", "conv_mm16_kernel", "conv_nr_kernel", "conv_d2_kernel", "conv32_d2p_kernel"), "f32": ("wino",),
             "f32x3": ("conv_x3_kernel", "wgrad_x3_kernel", "wgrad_x3s_kernel")}[dtype]
    assert "conv3x3_" in roof["kernel"] and any(n in roof["rocprof_kernel"] for n in names)
    assert roof["launches_per_step"] >= 1
    assert roof["avg_us"] > 0 and 0 < roof["share_of_step"] < 0.5 and roof["serial_step_us"] > 0
    kinds = {k.get("bound") for k in roof["other_kernels"]}
    assert "mfma" in kinds and all(k.get("frac", 0) <= 1.0 for k in roof["other_kernels"] if k.get("bound") in ("mfma", "hbm"))
    assert "value_f32_mfma" not in d and "value_h2" not in d       # (--dense-only: the named arithmetic only)
    rows = open(table).read().splitlines()
    assert rows[0].startswith("label,launches_per_step") and len(rows) > 25
    assert any("setmax_fwd" in ln and ",hbm," in ln for ln in rows)
    assert d["scaling"] == "weak" and d["n_gpus"] == 1 and d["config"]["distributed"] is None


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_other_fp32_class_lines_beside_the_headline(dev):
    """The bare command's `value` is the fp32-tensor arithmetic the product trains in (BASELINE configs[1..3] say fp32; the reference
    computes in fp32, nets/mj_uwyhNets_ba.py:428-462); the same job -- batch, weights, steps, warm-up -- is then timed on the Winograd
    fp32-MFMA kernels and in the f16x2 arithmetic and reported BESIDE it, never as `value`."""
    r = subprocess.run([sys.executable, BENCH, "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--clips-per-gpu", "8"],
                       env=_env(), capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["dtype"].startswith("f32") and "three-way bf16 split" in d["dtype"]
    assert d["dtype_f32_mfma"].startswith("f32") and "v_mfma_f32_16x16x4_f32" in d["dtype_f32_mfma"]
    assert "value_h2" not in d          # (the opt-in f16x2 set is timed only on request: --with-h2-line on a library built with --h2)
    for tag in ("f32_mfma",):
        assert d["value_" + tag] > 0 and abs(d["value_" + tag] * d["ms_per_step_" + tag] / 1e3 - 8) < 0.05
        assert abs(d["loss_" + tag] - d["loss"]) < 1e-3      # the same job: same batch, same initial weights, same number of steps
    rf = d["roofline_f32_mfma"]
    assert rf["bound"] == "mfma" and rf["peak"] == 157.3 and 0.05 < rf["frac"] <= 1.0 and "wino" in rf["rocprof_kernel"]
    assert d["value"] > 0 and "value_skip_masked" in d
    # the dense rate with every modality present (no constant placeholder frames), beside `value` and with its own dominant kernel
    assert d["value_all_present"] > 0 and abs(d["value_all_present"] * d["ms_per_step_all_present"] / 1e3 - 8) < 0.05
    ra = d["roofline_all_present"]
    assert ra["bound"] in ("mfma", "hbm") and 0.05 < ra["frac"] <= 1.0 and ra["stalled_launches"] == 0
    # the serialised pass is queued behind a gate (no event pair contains host time) and reports what it drops
    assert d["roofline"]["stalled_launches"] == 0 and d["roofline"]["gate"]["spin_ms"] >= d["roofline"]["gate"]["host_ms_per_profiled_step"]
