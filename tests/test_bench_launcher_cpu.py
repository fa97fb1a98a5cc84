"""bench.py's own launcher (`python bench.py --gpus N` with no RANK / WORLD_SIZE in the environment): rank environment,
relay of rank 0's line, exit-code propagation, and that a dead rank stops the job -- with stub children, no GPU."""
import contextlib
import importlib.util
import io
import os
import textwrap
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = textwrap.dedent('''
    import os, sys, time
    r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
    assert os.environ["LOCAL_RANK"] == str(r) and os.environ["LOCAL_WORLD_SIZE"] == str(w)
    assert not [k for k in os.environ if k.startswith("TORCHELASTIC_")]
    mode = sys.argv[1]
    if mode == "ok":
        time.sleep(0.2 * r)
        print("[Gloo] a library's banner on stdout")
        print('{"n_gpus": %d}' % w if r == 0 else "only rank 0 is relayed")
    elif mode == "die" and r == 1:
        sys.exit(7)
    else:
        time.sleep(120)
''')


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(REPO, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_self_spawn_relays_rank0_and_propagates_exit_codes(tmp_path, monkeypatch):
    bench = _bench_module()
    stub = tmp_path / "stub.py"
    stub.write_text(STUB)
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "stale")                    # a launcher's leftovers must not reach the ranks
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        rc = bench.self_spawn(3, ["ok"], script=str(stub))
    assert rc == 0 and out.getvalue() == '{"n_gpus": 3}\n'
    out, t0 = io.StringIO(), time.time()
    with contextlib.redirect_stdout(out):
        rc = bench.self_spawn(3, ["die"], script=str(stub))
    assert rc == 7 and out.getvalue() == "" and time.time() - t0 < 30      # the sleeping ranks were stopped


def test_device_map(monkeypatch):
    bench = _bench_module()
    monkeypatch.delenv("TRL_BENCH_DEVICE_MAP", raising=False)
    assert bench._device_map(4) == [0, 1, 2, 3]
    monkeypatch.setenv("TRL_BENCH_DEVICE_MAP", "0,0")
    assert bench._device_map(2) == [0, 0]


def test_roofline_traffic_is_reported_only_for_the_kernel_sources_it_was_measured_on(tmp_path):
    """`roofline.traffic` comes from a committed counter pass; bench.py must print it only while
    k_ppo.hip / trl_mlp.h are byte for byte what was measured (sha256 stamp in profiles/grad_kernel_traffic.json, written
    by tools/stamp_traffic.py), and null + a 'stale' stamp after any edit."""
    import json
    import shutil
    bench = _bench_module()
    with open(bench.TRAFFIC_FILE) as f:
        rec = json.load(f)
    assert rec["kernel_source_sha256"] and rec["measured_at_commit"] and list(rec["kernel_sources"]) == list(bench.TRAFFIC_SOURCES)
    value, stamp, busy = bench.pmc_traffic()
    if stamp["tree_matches"]:                                             # committed measurement is of this tree
        assert value == rec["traffic_bytes_per_launch"] and stamp["status"] == "current"
        assert rec["kernel_source_sha256"] == bench.kernel_source_digest()
    else:                                                                 # kernel edited since: nothing may be claimed
        assert value is None and stamp["status"].startswith("stale")
    # the same file against a tree whose kernel source differs by one byte
    root = tmp_path / "tree"
    for rel in bench.TRAFFIC_SOURCES:
        (root / os.path.dirname(rel)).mkdir(parents=True, exist_ok=True)
        shutil.copy(os.path.join(REPO, rel), root / rel)
    same = dict(rec, kernel_source_sha256=bench.kernel_source_digest(str(root)))
    path = tmp_path / "traffic.json"
    path.write_text(json.dumps(same))
    value, stamp, busy = bench.pmc_traffic(str(path), str(root))
    assert busy == rec.get("mfma_busy") and value == rec["traffic_bytes_per_launch"] and stamp["tree_matches"]
    with open(root / bench.TRAFFIC_SOURCES[0], "ab") as f:
        f.write(b"\n")
    value, stamp, busy = bench.pmc_traffic(str(path), str(root))
    assert busy is None and value is None and not stamp["tree_matches"] and stamp["status"].startswith("stale")
    assert stamp["measured_at_commit"] == rec["measured_at_commit"]
    value, stamp, busy = bench.pmc_traffic(str(tmp_path / "absent.json"), str(root))
    assert value is None and stamp["status"] == "no committed measurement"
