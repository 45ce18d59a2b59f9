"""bench.py's one-line JSON contract (driver side): run it small on the GPU and check the keys the driver and the judge read."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_has_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--size", "8000000", "--steps", "5", "--warmup", "2"], capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "MiB/s" and d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["config"]["bit_exact"] is True
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert d["value"] > 0 and d["ms_per_step"] > 0
    # the timed region is repeated in the process: `value` is the median region, the spread is reported
    m = d["mib_s"]
    assert m["repeats"] >= 15 and len(m["per_repeat"]) == m["repeats"] and m["min"] <= m["p10"] <= m["median"] <= m["p90"] <= m["max"]
    assert abs(m["median"] - d["value"]) < 1e-6 * d["value"]
    assert m["independent_streams_overlapped"]["value"] > 0 and d["single_wavefront_no_plan"]["value"] > 0  # all three figures
    one = m["independent_streams_one_launch"]  # K independent streams decoded by ONE launch, validated bit-exact inside bench.py
    assert one["streams"] == 4 and one["batch"]["launches"] == 1 and one["batch"]["direct_members"] == 4 and one["value"] > 0 and 0 < one["frac_of_hbm_peak"] < 1
    # counters are only ever reported for the kernels as they are now
    assert len(d["config"]["kernel_source_sha256"]) == 64
    assert rf["traffic"] is None or rf["traffic_stale"] is None


@pytest.mark.parametrize("n", (2, 4, 8))
def test_bench_spawns_its_own_ranks(n):
    """`python bench.py --gpus N` with no launcher above it starts N ranks itself and relays exactly ONE line (rehearsal mode: gloo
    ranks, sub-runs decoded by the library's host decoder — the spawn, sharding and pipelined-exchange logic without a GPU)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--rehearse", "--size", "2500000", "--block", "65536", "--interval", "32",
                        "--steps", "2", "--warmup", "1", "--parts", "3"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{")
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["config"]["n_ranks_seen"] == n and d["config"]["backend"] == "gloo" and d["scaling"] == "strong"
    assert "rehearsal" in d  # never mistaken for a measurement
    for k in ("none", "all", "all_unpipelined", "root", "root_unweighted_unpipelined", "one"):
        assert k in d["gather"], k
        assert len(d["gather"][k]["per_rank"]) == n
    assert d["gather"]["all"]["pipelined"] and d["gather"]["root"]["pipelined"] and not d["gather"]["all_unpipelined"]["pipelined"]
    assert sum(d["gather"]["root"]["shares"]) == 2500000 and d["gather"]["one"]["shares"][0] == 2500000
    assert 1 / n - 1e-9 <= d["config"]["root_share"] <= 1.0
    # a rank that is not the root of a root gather holds only its own range
    for pr in d["gather"]["root"]["per_rank"][1:]:
        assert pr["out_bytes_held"] == pr["range"][1] - pr["range"][0]


def test_bench_shards_a_block_stream_too():
    """BASELINE config 4 names the block_ container: the same sharded workload on a block_ stream (its chains come from the inline
    headers), three ranks over gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--rehearse", "--container", "block", "--size", "2000003", "--block", "65536",
                        "--interval", "32", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["config"]["container"] == "block_" and d["config"]["n_ranks_seen"] == 3 and d["config"]["bit_exact"]
    assert sum(d["gather"]["all"]["shares"]) == 2000003


def test_bench_passes_a_rank_failure_on():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    # --rehearse with the headline workload is refused by every rank: the launcher must exit non-zero and print no result line
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse", "--workload", "headline"], capture_output=True, text=True,
                       timeout=600, cwd=ROOT, env=env)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_host_workload_line():
    """`--workload host` (BASELINE config 5's shape on one GPU): its own metric, PCIe roofline, bit-exact."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "host", "--size", str(64 << 20), "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["config"]["bit_exact"] is True and d["roofline"]["bound"] == "pcie"
    assert "host memory" in d["metric"].lower() and d["value"] > 0
