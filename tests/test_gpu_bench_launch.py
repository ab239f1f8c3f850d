"""GPU (-m gpu): `python bench.py --gpus N` -- the driver's command shape -- must run N ranks (one process per GPU over
torch.distributed) and report n_gpus == N.  On a box with fewer GPUs than ranks the ranks share a device (gloo for the control
plane: RCCL refuses two ranks on one device), which exercises the whole multi-process path functionally; the line says so."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, env=env, cwd=ROOT,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                  # rank 0 prints ONE line
    return json.loads(lines[0])


def test_bench_gpus_2_launches_two_ranks_block_batches():
    out = _bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--blocks", "1024", "--no-cpu-baseline")
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak"
    assert out["verified"].startswith("round trip bit-exact")
    assert out["value"] > 0 and out["config"]["blocks_per_gpu"] == 1024


def test_bench_gpus_2_launches_two_ranks_sharded_frame():
    out = _bench("--gpus", "2", "--config", "4", "--steps", "1", "--warmup", "1", "--blocks", "6", "--no-cpu-baseline")
    assert out["n_gpus"] == 2
    assert "decoded by the oracle's FrameDecoder" in out["verified"]        # the gathered 2-rank frame through the reference's decoder


def test_bench_gpus_1_default_shape():
    out = _bench("--gpus", "1", "--steps", "2", "--warmup", "1", "--blocks", "2048")
    assert out["n_gpus"] == 1 and out["roofline"]["bound"] == "hbm" and out["roofline"]["frac"] > 0
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["gpu_blocks_decoded_by_oracle"] == 2048
