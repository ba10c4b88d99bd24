"""bench.py's launcher: `python bench.py --gpus N` by itself must start N ranks (torch.distributed.run, one process per GPU)
and fail -- if it has to -- only on the number of GPUs, never on how it was started.

CPU part (-m "not gpu"): without any GPU every rank stops at "needs a HIP device", after the launcher has spawned both.
GPU part (-m gpu, one MI355X): (a) `--gpus 2` reaches the device-count check of rank 1 and says so; (b) `--gpus 2 --share-gpu`
runs the whole two-rank harness on the one device (gloo carries the barriers, the mix queue reduces locally) and prints the one
JSON line with n_gpus = 2, per_gpu_efficiency and value_like_for_like_n1."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

ENV = dict(os.environ, OMP_NUM_THREADS="1")
ENV.pop("WORLD_SIZE", None)
ENV.pop("RANK", None)
ENV.pop("LOCAL_RANK", None)


def _run(args, timeout=600, env=None):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                          env=dict(ENV, **(env or {})), cwd=ROOT)


def test_self_launch_spawns_ranks_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: covered by the gpu tests below")
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert r.returncode != 0
    err = r.stderr + r.stdout
    assert "needs a HIP device" in err, err[-3000:]
    assert "launch with torch.distributed.run" not in err and "WORLD_SIZE=" not in err, err[-3000:]
    # both ranks were started (torch.distributed.run reports the failed ranks)
    assert "rank" in err.lower()


@pytest.mark.gpu
def test_two_ranks_on_a_one_gpu_box_fail_only_at_the_device_count():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this node has two GPUs")
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert r.returncode != 0
    err = r.stderr + r.stdout
    assert "--gpus 2 needs 2 GPUs" in err, err[-3000:]


@pytest.mark.gpu
def test_two_ranks_sharing_one_gpu_print_the_scaling_fields():
    r = _run(["--gpus", "2", "--share-gpu", "--steps", "64", "--warmup", "8", "--no-cpu-baseline"])
    assert r.returncode == 0, (r.stderr + r.stdout)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["share_gpu_test"] is True and d["scaling"] == "weak"
    assert 0.0 < d["per_gpu_efficiency"] <= 1.0
    assert d["value_like_for_like_n1"] > 0 and d["step_ms_without_reduce"] > 0
    assert d["roofline"]["kernel"] == "osc_mix_kernel"
    # BASELINE's multi-GPU config (configs[4]: granular time-stretch, streams sharded over the ranks, one stereo reduce per render)
    # rides along in every N > 1 line
    c5 = d["configs"]["config5"]
    assert "error" not in c5, c5
    assert c5["n_gpus"] == 2 and c5["value"] > 0 and c5["ms_per_step"] > 0 and c5["scaling"] == "weak"
    # ... and so does config 3 sharded the same way: K2f with the maxiMix::stereo mixdown fused (mxg_voice_render_mix_rows), the rows folded
    # and reduced through the grouped mix queue (round 6)
    c3 = d["configs"]["config3"]
    assert "error" not in c3, c3
    assert c3["n_gpus"] == 2 and c3["value"] > 0 and c3["ms_per_step"] > 0 and c3["scaling"] == "weak" and "fused" in c3["workload"]
    assert len(lines[0]) < 7000


@pytest.mark.gpu
def test_two_ranks_with_the_fallback_exchange():
    """bench.py's safety net for a multi-GPU node on which mxg_comm_create fails: the same step with torch tensors as staging and
    torch.distributed.reduce as the sum (maximilian_amd.dist.TorchMixQueue), forced here over gloo with two ranks on one GPU; the
    line says so."""
    r = _run(["--gpus", "2", "--share-gpu", "--steps", "64", "--warmup", "8", "--no-cpu-baseline"],
             env={"MXG_BENCH_FORCE_TORCH_EXCHANGE": "1"})
    assert r.returncode == 0, (r.stderr + r.stdout)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["exchange"].startswith("FALLBACK: torch.distributed.reduce"), d.get("exchange")
    assert d["value"] > 0 and 0.0 < d["per_gpu_efficiency"] <= 1.0


@pytest.mark.gpu
def test_default_line_carries_every_gpu_config():
    """The command the driver runs (`bench.py --gpus 1 --steps K --warmup W`) reports configs 3, 4, 4-mfma, 5 and the fused-mixdown
    step next to the headline: each with its own ms_per_step, value and the roofline of its dominant kernel."""
    r = _run(["--gpus", "1", "--steps", "20", "--warmup", "5"], timeout=1200)
    assert r.returncode == 0, (r.stderr + r.stdout)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    # the driver keeps an 8 KB tail of the line: the whole line has to fit; texts are said once (`notes`), --verbose has the rest
    assert len(lines[0]) < 7000, len(lines[0])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["roofline"]["kernel"] == "osc_kernel" and 0.3 < d["roofline"]["frac"] < 1.0
    assert d["notes"]["kernel_ms"] and 0.3 < d["roofline"]["frac_wall"] <= d["roofline"]["frac"] * 1.02
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["kind"] in ("reference", "port")
    want = {"config2_mixdown": "osc_mix_kernel", "config2_tables": "osctab_kernel", "config3": "voice_kernel", "config3_modB": "voice_kernel",
            "config3_mixdown": "voice_kernel", "sample_bank": "sample_parts_kernel",
            "config4": "fft_mfcc_kernel", "config4_walk": "fft_mfcc_kernel", "config4_mfma": "fft_mfcc_kernel", "config5": "granular_unit_kernel"}
    assert set(d["configs"]) == set(want), d["configs"].keys()
    for name, kernel in want.items():
        c = d["configs"][name]
        assert "error" not in c, c
        assert c["ms_per_step"] > 0 and c["value"] > 0 and c["roofline"]["kernel"] == kernel, c
        assert 0.0 < c["roofline"]["frac"] < 1.0 and c["roofline"]["kernel_ms"] <= c["ms_per_step"] * 1.001, c
        assert ("matrix_pipe" in c["roofline"]) == (name in ("config4_mfma", "config4_walk")), c["roofline"].keys()
        assert len(c["workload"]) <= 120
        if name != "config2_tables":  # (the per-voice-table extension has no reference CPU path to time)
            assert c["cpu_baseline"]["value"] > 0 and c["cpu_baseline"]["kind"] in ("reference", "port"), c.get("cpu_baseline")
    # the matrix-pipe form of config 4 is the library's default when only the coefficients are requested: the two entries measure the same kernel
    # (10-step runs a few seconds apart on a chip that has just been loaded differently: two measurements of the SAME kernel were
    # seen 5.6 % apart; these are regression guards, the claims are in profiles/r05_config4_ab.md from interleaved runs)
    assert abs(d["configs"]["config4"]["ms_per_step"] / d["configs"]["config4_mfma"]["ms_per_step"] - 1.0) < 0.15
    assert min(d["configs"]["config4_mfma"]["ms_per_step"], d["configs"]["config4"]["ms_per_step"]) <= d["configs"]["config4_walk"]["ms_per_step"] * 1.05
    # the fused-mixdown step (what every rank of an N > 1 run does per block) costs about what the plain render costs
    # (against the headline's GPU-side step time: at the driver's 20 steps the wall-clock figure carries ~3 us of fence per step)
    # (a regression guard, not the claim: measured 1.07-1.27 by box -- K1 39.9-44.6 us, the mixdown step 47.5-51.7; round 3: 1.32)
    assert d["configs"]["config2_mixdown"]["ms_per_step"] < 1.3 * d["step_ms_gpu"], (d["configs"]["config2_mixdown"], d["step_ms_gpu"])
    # config 3's N > 1 step on one GPU (K2f with the mixdown fused, producer / consumer wavefronts): VERDICT r05 #1 asks <= 1.15 x config 3's
    # own step (measured 1.10); the guard leaves room for a box's mood
    assert d["configs"]["config3_mixdown"]["step_vs_config3"] < 1.25, d["configs"]["config3_mixdown"]
