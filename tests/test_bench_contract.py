"""The bench line committed under profiles/ (the last default `python bench.py` run of the round on an MI355X) against the
driver's contract: required keys, the `roofline` and `cpu_baseline` objects, internal consistency, and the metric of
BASELINE.json.  A schema regression in bench.py shows up here on the CPU, not at round end."""

import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    with open(os.path.join(ROOT, "profiles", "r03", name)) as fh:
        rows = [l for l in fh.read().splitlines() if l.startswith("{")]
    assert len(rows) == 1, "exactly one JSON line"
    return json.loads(rows[0])


def test_default_line_follows_the_contract():
    d = _line("bench_n1.json")
    with open(os.path.join(ROOT, "BASELINE.json")) as fh:
        base = json.load(fh)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "frames/s" and base["metric"].startswith(d["metric"])      # BASELINE.json's headline metric
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["scaling"] in ("weak", "strong")
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = frames of the job / step time
    frames = d["config"]["frames_total"]
    assert abs(d["value"] - frames / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # achieved = algorithmic bytes (24 N + 72 per frame) / the live launch duration
    n = d["config"]["n_atoms"]
    assert r["algorithmic_bytes"] == r["frames_in_launch"] * (24 * n + 72)
    assert abs(r["achieved"] - r["algorithmic_bytes"] / r["launch_seconds"] / 1e9) / r["achieved"] < 1e-9
    assert r["launch_seconds"] * 1e3 <= d["ms_per_step"]            # the dominant kernel fits inside a step
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["unit"] == d["unit"]
    assert d["verified"] is True and d["verification"]["ok"] is True


def test_two_rank_rehearsal_line():
    d = _line("bench_2rank_rehearsal.json")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["verified"] is True
    assert d["config"]["trajectory_identical_on_all_ranks"] is True
    ranks = d["per_rank"]
    assert [r["rank"] for r in ranks] == [0, 1]
    # the two ranks' frame shards tile the trajectory
    assert ranks[0]["rdf_frames"][0] == 0 and ranks[0]["rdf_frames"][1] == ranks[1]["rdf_frames"][0]
    assert ranks[1]["rdf_frames"][1] == d["config"]["frames_total"]


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` from a plain shell (the shape of the driver's command): the parent starts the ranks
    as a child process through torch.distributed.run on 127.0.0.1, relays their output and their exit code.  Checked
    here without a GPU through --rank-probe (every rank reports RANK / WORLD_SIZE and exits before touching torch)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    bench = os.path.join(ROOT, "bench.py")
    ok = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--rank-probe", "0"], env=env,
                        capture_output=True, text=True, timeout=300)
    assert ok.returncode == 0, ok.stderr[-2000:]
    probes = [json.loads(l) for l in ok.stdout.splitlines() if l.startswith("{")]
    assert sorted(p["rank"] for p in probes) == [0, 1]
    assert all(p["world"] == 2 and p["gpus"] == 2 and p["master_addr"] == "127.0.0.1" for p in probes)
    bad = subprocess.run([sys.executable, bench, "--gpus", "2", "--rank-probe", "3"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert bad.returncode != 0                      # a failing rank is not swallowed by the launcher
    # under a launcher with a different world size the mismatch is an error, not a silent single-rank run
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    one = subprocess.run([sys.executable, bench, "--gpus", "1", "--rank-probe", "0"], env=env2, capture_output=True,
                         text=True, timeout=300)
    assert one.returncode == 0 and json.loads(one.stdout.strip())["world"] == 1
