"""CPU: what `python bench.py --gpus N` would start -- one process per GPU with the variables a launcher sets -- and the
presets of the bench's configurations (no GPU, nothing is started)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_rank_commands_for_eight_gpus():
    import bench
    argv = ["--gpus", "8", "--steps", "5", "--warmup", "1", "--config", "3"]
    cmds = bench.rank_commands(8, argv, 29555)
    assert len(cmds) == 8
    for r, (cmd, env) in enumerate(cmds):
        assert cmd[0] == sys.executable and cmd[1] == os.path.join(ROOT, "bench.py") and cmd[2:] == argv
        assert env == {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": "8", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29555"}
    assert len({env["LOCAL_RANK"] for _, env in cmds}) == 8          # one GPU each


@pytest.mark.parametrize("world,want", [(1, 100_000_000), (2, 50_000_000), (4, 25_000_000), (8, 12_500_000)])
def test_config3_shards_a_fixed_total(monkeypatch, world, want):
    import bench
    monkeypatch.setenv("WORLD_SIZE", str(world))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", str(world), "--config", "3"])
    a = bench.parse_args()
    assert a.pairs == want and a.genome == "grch38" and a.introns == 300000 and not a.pairs_given
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", str(world), "--config", "3", "--pairs", "12500000"])
    a = bench.parse_args()
    assert a.pairs == 12_500_000 and a.pairs_given


def test_default_is_config2(monkeypatch):
    import bench
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse_args()
    assert a.pairs == 10_000_000 and a.genome == "chr20" and a.config == 2 and a.gpus == 1 and a.multihit_frac == 0.05
