"""bench.py's own launcher (`python bench.py --gpus N` with no launcher environment): N fresh rank processes with the
rank environment a launcher would set, the worst exit code relayed.  The ranks here are a stub script, not the bench
(no GPU needed)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_spawn_ranks_sets_the_rank_environment_and_relays_the_worst_exit_code(tmp_path, monkeypatch):
    import bench
    stub = tmp_path / "rank_stub.py"
    stub.write_text(
        "import json, os, sys\n"
        "env = {k: os.environ.get(k) for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}\n"
        "open(os.path.join(sys.argv[1], 'rank%s.json' % env['RANK']), 'w').write(json.dumps({'env': env, 'argv': sys.argv[2:]}))\n"
        "sys.exit(int(sys.argv[2]) if env['RANK'] == '2' else 0)\n")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    rc = bench.spawn_ranks(3, script=str(stub), argv=[str(tmp_path), "7", "--gpus", "3"])
    assert rc == 7                                      # rank 2 failed with 7: the launcher's own exit code
    seen = [json.loads((tmp_path / f"rank{r}.json").read_text()) for r in range(3)]
    assert [s["env"]["RANK"] for s in seen] == ["0", "1", "2"] and [s["env"]["LOCAL_RANK"] for s in seen] == ["0", "1", "2"]
    assert all(s["env"]["WORLD_SIZE"] == "3" and s["env"]["MASTER_ADDR"] == "127.0.0.1" for s in seen)
    assert len({s["env"]["MASTER_PORT"] for s in seen}) == 1 and seen[0]["env"]["MASTER_PORT"].isdigit()
    assert all(s["argv"] == ["7", "--gpus", "3"] for s in seen)
    assert bench.spawn_ranks(2, script=str(stub), argv=[str(tmp_path), "0"]) == 0


def test_spawn_ranks_takes_the_other_ranks_down_when_one_fails(tmp_path, monkeypatch):
    """One rank exits 5 at once while the others would sit in a rendezvous for minutes: the launcher must stop them and
    come back promptly with a failure code."""
    import time

    import bench
    stub = tmp_path / "rank_stub.py"
    stub.write_text(
        "import os, sys, time\n"
        "if os.environ['RANK'] == '1':\n"
        "    sys.exit(5)\n"
        "time.sleep(120)\n")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    t0 = time.monotonic()
    rc = bench.spawn_ranks(3, script=str(stub), argv=[])
    assert time.monotonic() - t0 < 30.0
    assert rc != 0 and rc >= 5
