"""GPU test of the C++ host side (include/expann/*.h + expann_amd/host/expann_bench): the CRTP
engine driven by the basic_bench counterpart with the reference's config keys
(config_synthetic.json:1-9), serial query_k and the batched extension."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "expann_amd", "host", "expann_bench")


def test_driver_config_keys_and_result_fields(tmp_path):
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(EXE)])
    cfg = tmp_path / "config_synthetic.json"
    # BASELINE config C1 shape with the reference's key names
    cfg.write_text(json.dumps({"dataset": "Synthetic", "ds_name": "c1", "num_threads": 1,
                               "n": 10000, "m": 100, "d": 64, "k": 10}))
    out = subprocess.run([EXE, "--config", str(cfg), "--m", "120"], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr
    lines = [json.loads(x) for x in out.stdout.strip().splitlines()]
    assert lines[0]["m"] == 120 and lines[0]["n"] == 10000     # CLI overrides the config file
    for bd in lines[1:]:
        for key in ("time_per_query_ns", "time_to_build_ns", "average_distance",
                    "average_squared_distance", "recall", "engine_name", "param_list"):
            assert key in bd                                     # src/bench_data.h:20-28
        assert bd["recall"] == 1.0                               # exact engine vs exact ground truth
        assert bd["time_per_query_ns"] > 0 and bd["average_squared_distance"] > 0
    assert {bd["param_list"]["mode"] for bd in lines[1:]} == {"serial", "batched"}


def test_driver_rejects_missing_parameters(tmp_path):
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(EXE)])
    out = subprocess.run([EXE, "--n", "100"], capture_output=True, text=True, cwd=tmp_path)
    assert out.returncode == 2 and "missing parameter" in out.stderr
