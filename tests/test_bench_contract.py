"""The bench line the driver parses (the committed run of the current build, profiles/): every field of the contract."""
import glob
import json
import os


def _latest():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    files = sorted(glob.glob(os.path.join(root, "r[0-9][0-9]_bench_steps20*.json")))       # newest round's line at the driver's flags
    assert files, "no committed bench line under profiles/"
    return json.load(open(files[-1]))


def test_bench_line_has_the_contract_fields():
    j = _latest()
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(j[k], t), (k, type(j[k]))
    assert j["vs_baseline"] is None                      # BASELINE.md publishes no number for this metric
    assert j["metric"] == "speech-tokens/sec" and j["higher_is_better"] is True and j["data"] == "synthetic"
    assert j["scaling"] in ("weak", "strong") and "workload" in j["config"] and "model" in j["config"]
    assert abs(j["value"] - 64 * 1e3 / j["ms_per_step"]) / j["value"] < 0.02      # whole-job tokens / wall time
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    assert r["traffic"] is None or r["traffic"] > r["bytes_per_step"]["weights_backbone"]
    assert abs(r["achieved"] * 1e9 * r["event_ms_per_step"] * 1e-3 - r["bytes_per_step"]["total"]) / r["bytes_per_step"]["total"] < 1e-6
    c = j["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
    assert c["unit"] == j["unit"]


def test_bench_gpus_n_launches_n_ranks_or_fails_loudly():
    """VERDICT r1 #4a / ADVICE r1: `python bench.py --gpus N` starts N ranks itself (children of a parent that has made no
    GPU call) and never reports a 1-rank number under N; a WORLD_SIZE that disagrees with --gpus is refused.  No GPU here:
    the children die at torch.cuda.set_device and the parent must exit non-zero without printing a JSON line."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert "self-launch:" in r.stderr and "--nproc-per-node=2" in r.stderr and "torch.distributed.run" in r.stderr
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 2 and "refusing" in r.stderr and not r.stdout.strip()


def test_bench_gpus_n_relays_the_childs_json_line(tmp_path):
    """when an 8-GPU node appears the first run must not be the first debug session -- the parent's relay of a
    SUCCESSFUL N-rank child is exercised here with a stub launcher in torch.distributed.run's place: it gets the launcher's
    arguments (one rank per GPU, 127.0.0.1 rendezvous, the bench's own flags), writes library noise and one JSON line to stdout;
    the parent prints exactly that line and exits 0."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    stub = tmp_path / "stub_launcher.py"
    stub.write_text(
        "import json, sys\n"
        "a = sys.argv[1:]\n"
        "assert '--nproc-per-node=4' in a and '--nnodes=1' in a and a[a.index('--master-addr') + 1] == '127.0.0.1', a\n"
        "i = [k for k, x in enumerate(a) if x.endswith('bench.py')][0]\n"
        "assert a[i + 1:] == ['--gpus', '4', '--steps', '3', '--warmup', '1'], a[i + 1:]\n"
        "print('RCCL version 2.x banner on stdout')\n"
        "print(json.dumps({'metric': 'speech-tokens/sec', 'value': 123.0, 'n_gpus': 4, 'steps': 3, 'warmup': 1}))\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(OMNI_BENCH_LAUNCHER="stub_launcher", PYTHONPATH=str(tmp_path) + os.pathsep + env.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1 and json.loads(lines[0]) == {"metric": "speech-tokens/sec", "value": 123.0, "n_gpus": 4, "steps": 3, "warmup": 1}
    # a child that prints a line but FAILS is not relayed
    stub.write_text("import sys\nprint('{\"value\": 1}')\nsys.exit(7)\n")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 7 and not r.stdout.strip()
