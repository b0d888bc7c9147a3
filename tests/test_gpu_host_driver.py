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


def test_driver_devices_flag_shards_the_engine(tmp_path):
    """--devices a,b,c: gpu_brute_force_engine::config{devices} -> expann_sharded_* (here three shards
    on the box's one GPU); the C++ job machinery sees the same engine interface, recall stays 1.0."""
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(EXE)])
    out = subprocess.run([EXE, "--dataset", "Synthetic", "--n", "30000", "--m", "150", "--d", "128", "--k", "10",
                          "--devices", "0,0,0", "--mode", "both"], capture_output=True, text=True,
                         timeout=300, cwd=tmp_path)
    assert out.returncode == 0, out.stderr
    lines = [json.loads(x) for x in out.stdout.strip().splitlines()]
    assert len(lines) == 3
    for bd in lines[1:]:
        assert bd["recall"] == 1.0
        assert bd["param_list"]["devices"] == "0,0,0" and bd["param_list"]["shards"] == "3"


def test_driver_rejects_missing_parameters(tmp_path):
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(EXE)])
    out = subprocess.run([EXE, "--n", "100"], capture_output=True, text=True, cwd=tmp_path)
    assert out.returncode == 2 and "missing parameter" in out.stderr
    out = subprocess.run([EXE, "--dataset", "Nope"], capture_output=True, text=True, cwd=tmp_path)
    assert out.returncode == 1 and "Invalid dataset type" in out.stderr


def _write_vecs(path, arr):
    """fvecs/ivecs: per vector a 32-bit d then d 4-byte components (src/dataset_loader.h:96-125)."""
    import numpy as np
    n, d = arr.shape
    rec = np.empty((n, d + 1), dtype=np.int32)
    rec[:, 0] = d
    rec[:, 1:] = arr.view(np.int32)
    rec.tofile(path)


def test_driver_sift1m_files_and_result_files(tmp_path, oracle):
    """--dataset Sift1M: fvecs/ivecs files in the reference's layout, ground truth from the ivecs
    file (here: the oracle's exact answers), results appended to data/<ds_name>/data/all.json and
    written to latest.json like bench_data_manager::save (src/bench_data_manager.h:65-73)."""
    import numpy as np
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(EXE)])
    rng = np.random.RandomState(3)
    base = np.clip(np.round(np.abs(rng.standard_normal((3000, 128))) * 40), 0, 255).astype(np.float32)
    query = np.clip(np.round(np.abs(rng.standard_normal((25, 128))) * 40), 0, 255).astype(np.float32)
    gt, _ = oracle.brute_force(base, query, 100)
    sift = tmp_path / "sift"
    sift.mkdir()
    _write_vecs(sift / "sift_base.fvecs", base)
    _write_vecs(sift / "sift_query.fvecs", query)
    _write_vecs(sift / "sift_groundtruth.ivecs", gt.astype(np.int32))
    args = [EXE, "--dataset", "Sift1M", "--ds_name", "mini_sift", "--num_threads", "1", "--k", "10",
            "--sift_dir", str(sift), "--save", "1", "--data_root", str(tmp_path / "data") + "/",
            "--mode", "batched"]
    for _ in range(2):
        out = subprocess.run(args, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
    bd = json.loads(out.stdout.strip().splitlines()[-1])
    assert bd["recall"] == 1.0          # exact engine vs the file's exact ground truth
    latest = json.load(open(tmp_path / "data" / "mini_sift" / "data" / "latest.json"))
    allj = json.load(open(tmp_path / "data" / "mini_sift" / "data" / "all.json"))
    assert len(latest) == 1 and len(allj) == 2 and allj[0]["engine_name"] == latest[0]["engine_name"]
    assert set(latest[0]) == {"time_per_query_ns", "time_to_build_ns", "average_distance",
                              "average_squared_distance", "recall", "engine_name", "param_list"}
