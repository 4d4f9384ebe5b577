"""bench.py's contract on a GPU box: one JSON line with the driver's keys, `--gpus N` launching N ranks by itself and
partitioning ONE test set over them (strong scaling), N = 1 carrying roofline + cpu_baseline + the WER leg."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_are_launched_and_share_one_test_set():
    d = run_bench("--workload", "tiny", "--gpus", "2", "--dist-backend", "gloo", "--device", "0", "--steps", "2", "--warmup", "1")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 2 and d["warmup"] == 1
    assert len(d["rank_wall_s"]) == 2
    assert d["config"]["utterances"] == 12 and d["config"]["utterances_rank0"] == 6       # LPT split of the one set
    assert d["value"] > 0 and d["unit"] == "audio-sec/wall-sec" and d["higher_is_better"] is True
    assert d["cpu_baseline"] is None                                                       # rank 0 at N = 1 only


def test_single_rank_line_has_roofline_cpu_baseline_and_wer():
    d = run_bench("--workload", "tiny", "--steps", "2", "--warmup", "1", "--cpu-budget", "2", "--wer-utts", "8")
    assert d["n_gpus"] == 1 and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"] and k in d["roofline_other_stage"]
    assert {d["roofline"]["bound"], d["roofline_other_stage"]["bound"]} == {"hbm", "mfma"}
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert cb["one_best_vs_cpu_decoder_same_loglikes"]["errors"] == 0
    assert d["wer"]["identical_wer_lines"] is True and d["wer"]["wer_line_device"].startswith("%WER")
    assert d["stage_ms"]["total_wall"] >= d["stage_ms"]["decode_queue_kernel"]
    assert d["decoder"]["failed_utterances"] == 0
