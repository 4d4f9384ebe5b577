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


@pytest.mark.parametrize("headline", ["random", "faithful"])
def test_four_ranks_search_their_longest_utterances_beside_the_model(headline):
    """Ranks of 4 and more get the long-utterance decoder (kamd_batch_decoder_set_long_decoder, 32 lanes): with the upload
    inside the timed region the 32 longest utterances of a rank's shard are stored first and searched beside the model of the
    others; no utterance may fail and the line keeps its shape."""
    d = run_bench("--workload", "tiny", "--utts", "640", "--gpus", "4", "--dist-backend", "gloo", "--device", "0", "--steps", "2", "--warmup", "1",
                  "--host-threads", "2", "--tokens-per-frame", "4000", "--headline", headline)        # (four ranks' arenas on ONE device here)
    assert d["n_gpus"] == 4 and len(d["rank_wall_s"]) == 4 and d["value"] > 0
    assert d["config"]["utterances"] == 640 and d["config"]["utterances_rank0"] == 160
    assert d["config"]["long_utterances_rank0"] == 32 and d["config"]["upload_in_timed_region"] is True
    assert d["decoder"]["failed_utterances"] == 0


def test_single_rank_line_has_roofline_cpu_baseline_and_wer():
    """The default line: `value` is the recipe-faithful configuration (i-vector model, planted transcripts)."""
    d = run_bench("--workload", "tiny", "--steps", "2", "--warmup", "1", "--cpu-budget", "2", "--wer-utts", "8")
    assert d["n_gpus"] == 1 and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["config"]["headline"] == "faithful" and "online_ivectors" in d["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"] and k in d["roofline_other_stage"]
    assert {d["roofline"]["bound"], d["roofline_other_stage"]["bound"]} == {"hbm", "mfma"}
    hbm = d["roofline"] if d["roofline"]["bound"] == "hbm" else d["roofline_other_stage"]
    for k in ("counter_frac", "wait_fraction", "write_amplification", "arcs_per_expanded_token"):
        assert k in hbm
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb and cb["single_thread"]["value"] > 0
    assert cb["one_best_vs_cpu_decoder_same_loglikes"]["errors"] == 0 and cb["one_best_vs_cpu_decoder_same_loglikes"]["ref_words"] > 0
    assert d["wer"]["identical_wer_lines"] is True and d["wer"]["wer_line_device"].startswith("%WER")
    assert d["stage_ms"]["total_wall"] >= d["stage_ms"]["decode_queue_kernel"] and d["stage_ms"]["ivector_extraction"] > 0
    assert d["decoder"]["failed_utterances"] == 0 and d["decoder"]["failures"] == []
    t = d["transcripts"]
    assert "error" not in t and t["wer_line"].startswith("%WER") and t["words_per_utterance"] > 1.5
    assert d["config"]["upload_in_timed_region"] is True and d["upload"]["bytes"] > 0 and d["upload"]["passes"] >= 1
    assert d["hbm_resident_value"] > 0
    assert "sgemm" in cb["nnet"] and cb["nnet_only_per_core"]["scalar_oracle"] > 0
    assert d["roofline_other_stage"]["flops_per_step"] > 0 or d["roofline"].get("flops_per_step", 0) > 0
    r = d["random_loglikes"]
    assert "error" not in r and r["value"] > 0 and r["roofline"]["bound"] == "hbm" and r["decoder"]["failed_utterances"] == 0
    p = d["planted"]
    assert "error" not in p and p["wer_line"].startswith("%WER") and p["failed_utterances"] == 0
    assert "error" not in d["online_ivectors"] and d["online_ivectors"]["value"] > 0
    sl = d["streaming"]
    assert "error" not in sl and sl["ms_per_chunk"] > 0 and sl["aggregate_x_rt"] > 0 and sl["finalize_ms"] > 0 and sl["ms_per_tick_256"] > 0


def test_random_headline_keeps_round_3_legs():
    d = run_bench("--workload", "tiny", "--steps", "2", "--warmup", "1", "--headline", "random", "--no-cpu-baseline", "--no-wer", "--no-streaming")
    assert d["config"]["headline"] == "random" and d["decoder"]["failed_utterances"] == 0
    p = d["planted"]
    assert "error" not in p and p["wer_line"].startswith("%WER") and p["words_per_utterance"] > 1.5
    assert p["determinized_lattice_depth"] >= 1.0 and p["host_tail_cpu_ms_per_utterance"] > 0
    assert "error" not in d["online_ivectors"] and d["online_ivectors"]["value"] > 0


def test_resident_flag_keeps_round_2_contract():
    d = run_bench("--workload", "tiny", "--steps", "1", "--warmup", "1", "--resident", "--no-cpu-baseline", "--no-wer", "--no-random-leg", "--no-streaming",
                  "--no-planted", "--no-ivector-leg")
    assert d["config"]["upload_in_timed_region"] is False and d["upload"] is None and "hbm_resident_value" not in d


def test_rccl_code_path_with_one_rank():
    """What a one-GPU box can say about the RCCL path: `--force-dist --dist-backend nccl` runs the N > 1
    code of bench.py with ONE rank -- torch.cuda.set_device beside kamd_set_device, the RCCL communicator created in the
    process that holds the library's streams and page-locked pools, the barriers around the timed region, the all-gather of
    the rank walls and the scalar all-reduce on device tensors -- and the line keeps its N = 1 shape."""
    d = run_bench("--workload", "tiny", "--gpus", "1", "--force-dist", "--dist-backend", "nccl", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                  "--no-wer", "--no-streaming", "--no-random-leg", "--no-planted", "--no-ivector-leg")
    assert d["n_gpus"] == 1 and len(d["rank_wall_s"]) == 1 and d["value"] > 0
    assert d["config"]["utterances"] == d["config"]["utterances_rank0"] == 12
    assert abs(d["rank_wall_s"][0] - d["rank0_wall_s"]) < 1e-9                    # (the gathered wall is this rank's own)


def test_rccl_ranks_when_the_box_has_more_than_one_gpu():
    """The N > 1 path over RCCL (--dist-backend nccl, one rank per GPU): runs wherever two devices are visible; the pool's
    one-GPU boxes skip it (the gloo test above covers the same code path with two ranks on one device)."""
    from kaldi_amd._lib import lib
    n = lib().kamd_device_count()
    if n < 2:
        pytest.skip("one GPU visible")
    g = min(n, 8)
    d = run_bench("--workload", "tiny", "--utts", str(6 * g), "--gpus", str(g), "--dist-backend", "nccl", "--steps", "2", "--warmup", "1")
    assert d["n_gpus"] == g and len(d["rank_wall_s"]) == g and d["value"] > 0
    assert d["config"]["utterances"] == 6 * g and d["config"]["utterances_rank0"] == 6
