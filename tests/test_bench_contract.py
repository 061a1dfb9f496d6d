"""bench.py / __graft_entry__ contracts on a real GPU: one JSON line with the agreed keys, smoke() green."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run_bench(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "60", "--warmup", "10", *extra],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-2000:]   # stdout = the result line and nothing else
    return json.loads(lines[0])


def test_bench_prints_one_json_line_with_the_contract_keys():
    d = run_bench()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 60 and d["warmup"] == 10 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic" and d["scaling"] == "weak"
    assert "Panda 7-DoF K=4096 H=20" in d["metric"] and "workload" in d["config"]
    assert d["value"] > 100.0                                   # the BASELINE target (Hz) with a very wide margin
    assert d["value"] == pytest.approx(1e3 / d["ms_per_step"], rel=1e-6)
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "issue" and r["unit"] == "GB/s" and r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and r["hbm_frac"] == r["frac"]
    assert 0.05 < r["kernel_ms"] < 5.0
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample", "rows"):
        assert key in c, key
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    assert [row["threads"] for row in c["rows"]] == [1, c["cores"], 1, c["cores"]] and c["value"] == c["rows"][1]["value"]
    lat = d["latency_ms"]
    assert lat["p5"] <= lat["median"] <= lat["p95"] and lat["median"] == pytest.approx(d["ms_per_step"], rel=0.25)
    i = r["issue"]                                              # instruction-issue view from the committed SQ counters
    assert i is None or (0 < i["frac_of_fp32_issue_peak"] < 1 and i["waves"] == r["wavefronts"])   # the committed SQ counters belong to the kernel layout that ran
    assert d["config"]["final_ee_to_goal_m"] < 0.6              # the closed loop moves towards the goal
    # the reference-API loop (MPPIisaacPlanner.compute_action_tensor with torch.save blobs + a Python-stepped K = 1 world) is on the line
    assert d["value_facade"] > 1000.0 and d["value_generic_objective"] > 300.0 and d["value_facade"] <= 1.05 * d["value"]
    f = d["config"]["facade"]
    assert f["fused"]["final_ee_to_goal_m"] < 0.6 and f["generic"]["final_ee_to_goal_m"] < 0.6 and f["generic_graph_safe"]["value"] >= 0.8 * f["generic_untraced"]["value"]
    # round 6: the reference-style Objective of the `generic` row (nothing declared) is traced into the in-kernel cost and runs at the
    # fused rate, validations included; the rows say which path ran
    assert f["generic"]["mode"].startswith("traced -> in-kernel") and f["generic_untraced"]["mode"].startswith("generic")
    assert d["value_generic_objective"] >= 0.7 * d["value_facade"] and d["value_generic_objective_untraced"] == f["generic_untraced"]["value"]


def test_bench_other_workloads_and_the_sharded_code_path():
    d = run_bench("--workload", "boxer_push", "--no-cpu-baseline")
    assert "boxer_push" in d["metric"] and d["roofline"]["kernel"] == "k_rollout_scene_quad" and d["value"] > 10
    # the contact workloads' lines carry their task outcome, and what the numbers mean: the goal of the pushing scene lies inside an
    # obstacle's footprint (the block is done when it rests against it)
    o = d["config"]["task_outcome"]
    assert o["goal_is_inside_the_footprint_of"] == "paper_obst1" and 0.3 < o["closest_the_block_can_get_m"] < 0.6
    assert 0.0 < o["final_block_to_goal_m"] < 2.0 and o["iterations"] > 0
    env = dict(os.environ, MPPI_BENCH_FORCE_DIST="1", MASTER_PORT="29547")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "60", "--warmup", "10", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 1, out.stdout[-2000:]                 # (RCCL's version banner goes to stderr)
    d = json.loads(lines[0])
    assert "sample-shard x1" in d["config"]["parallelism"] and "nccl" in d["config"]["parallelism"] and d["value"] > 100.0   # RCCL process group + all-gather, one rank


def test_bench_two_ranks_launched_like_the_driver_does():
    """`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`: env rendezvous on 127.0.0.1, sample shards,
    record all-gather, max-over-ranks timing, rank 0 prints.  On this one-GPU box the two ranks share the GPU and the
    exchange goes over gloo (MPPI_BENCH_BACKEND); on the 8-GPU node the same code runs over RCCL."""
    env = dict(os.environ, MPPI_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29613", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "100", "--warmup", "10"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 1, out.stdout[-2000:]                 # rank 0 only, no library banners
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "cpu_baseline" not in d
    assert d["config"]["K_per_gpu"] == 4096 and d["config"]["K_total"] == 8192
    assert d["value"] == pytest.approx(2 * d["config"]["loop_hz"], rel=1e-6) and d["value"] > 100.0
    assert d["config"]["final_ee_to_goal_m"] < 0.6


def test_plain_python_bench_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (a driver that does not go through torch.distributed.run):
    bench.py becomes the launcher - two ranks on 127.0.0.1, ONE valid result line on stdout"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MPPI_BENCH_BACKEND="gloo", MPPI_BENCH_SECOND="0", MPPI_BENCH_SHIPPED="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "60", "--warmup", "10"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["K_total"] == 8192 and d["value"] > 100.0
    per_rank = d["config"]["exchange"]["per_rank"]
    assert [r["dist_world_size"] for r in per_rank] == [2, 2] and [r["dist_rank"] for r in per_rank] == [0, 1]


def test_plain_python_bench_gpus_8_as_the_driver_will_run_it():
    """the driver's 8-GPU command, `python bench.py --gpus 8 --steps 20 --warmup 5`, before there is an 8-GPU node to run it on: eight
    ranks launched by bench.py itself, here all on the one device (gloo staging, MPPI_BENCH_BACKEND) - ONE valid line, the weak row
    at 8 x 4096 samples, BASELINE config 5 at its stated size as the strong row (65536 samples = 8192 per rank, H = 30), a report
    from every rank, and the mailbox A/B pass that follows the result line (forced here: it needs no RCCL) leaves the exit code alone"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MPPI_BENCH_BACKEND="gloo", MPPI_BENCH_MAILBOX_AB="force", MPPI_BENCH_SHIPPED="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"],
                         capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["config"]["K_per_gpu"] == 4096 and d["config"]["K_total"] == 32768
    assert d["value"] == pytest.approx(8 * d["config"]["loop_hz"], rel=1e-6) and d["value"] > 100.0
    s5 = d["config"]["cfg5_strong"]
    assert s5 is not None and s5["K_total"] == 65536 and s5["K_per_gpu"] == 8192 and s5["H"] == 30 and s5["loop_hz"] > 1.0
    per_rank = d["config"]["exchange"]["per_rank"]
    assert [r["dist_rank"] for r in per_rank] == list(range(8)) and all(r["dist_world_size"] == 8 for r in per_rank)
    assert all("peer_access" in r for r in per_rank)
    ab = [l for l in out.stderr.splitlines() if l.startswith("[bench] mailbox_ab ")]
    assert len(ab) == 1, out.stderr[-3000:]
    ab = json.loads(ab[0][len("[bench] mailbox_ab "):])
    assert ab["n_gpus"] == 8 and len(ab["per_rank"]) == 8 and ab["selected"] in ("mailbox", "rccl") and ab["why"]


def test_bench_two_ranks_exchange_through_the_library_mailbox():
    """the same two-rank launch with the library's own record exchange (mppi_mailbox_*): the two processes connect their inboxes
    through hipIpc handles, bench.py's probe compares three iterations of mailbox-gathered records with the all-gather bit for
    bit, and the timed loop then runs without a collective (and as a captured graph).  Here both ranks share the one GPU; on
    the 8-GPU node the inboxes are peers over xGMI."""
    env = dict(os.environ, MPPI_BENCH_BACKEND="gloo", MPPI_BENCH_EXCHANGE="mailbox", MPPI_BENCH_SECOND="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29617", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "100", "--warmup", "10"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "mailbox exchange" in d["config"]["parallelism"], (d["config"]["parallelism"], out.stderr[-1500:])
    assert d["value"] > 100.0 and d["config"]["final_ee_to_goal_m"] < 0.6
    assert d["config"]["exchange_ms"] is not None and d["config"]["exchange_ms"] < 5.0
    ex = d["config"]["exchange"]                                      # the record explains itself: what ran on every rank, and why
    assert ex["selected"] == "mailbox" and "probe passed" in ex["why"]
    assert [r["selected"] for r in ex["per_rank"]] == ["mailbox", "mailbox"] and all(r["mppi_exchange_status"] == 0 for r in ex["per_rank"])
    assert all(r["probe"]["equal_per_rank"] == [1, 1] and r["probe"]["late"] == [0, 0] for r in ex["per_rank"])


def test_bench_a_rank_that_refuses_the_mailbox_sends_every_rank_to_the_all_gather():
    """negative path of the multi-rank set-up: one rank cannot create its inbox (test hook) - every rank agrees on the all-gather,
    nobody hangs, and the result line says which exchange ran and why, per rank"""
    env = dict(os.environ, MPPI_BENCH_BACKEND="gloo", MPPI_BENCH_EXCHANGE="mailbox", MPPI_BENCH_SECOND="0", MPPI_BENCH_SHIPPED="0",
               MPPI_BENCH_TEST_REFUSE_RANK="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29619", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "50", "--warmup", "5"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    ex = d["config"]["exchange"]
    assert ex["selected"] == "rccl" and "refused" in ex["why"] and "rank 1" in ex["why"]
    assert [r["selected"] for r in ex["per_rank"]] == ["rccl", "rccl"]
    assert ex["exchange_ms"] is None or ex["exchange_ms"] >= 0
    assert "all-gather" in d["config"]["parallelism"] and d["value"] > 100.0


def test_smoke_entry_point():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.smoke()
