"""bench.py's stdout contract: the LAST line is one compact JSON object the driver can parse (round 3's 23-KB line was not parsed)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _full_record():
    big = {f"k_some_kernel<{i}, true, 128>": 1.2345678901 * i for i in range(400)}
    step = {"flops_per_step": 2.94e12, "gemm_flops_per_step": 1.78e12, "message_valu_flops_per_step": 1.16e12, "gemm_engine": "split-bf16", "fp32_bound_ms": 18.7,
            "split_engine_bound_ms": 11.6, "hbm_bound_ms": 4.6, "binding_roof": "arithmetic", "achieved_TFLOPs": 54.3, "frac_of_binding_roof": 0.21,
            "frac_of_hbm_roof": 0.085, "note": "x" * 3000}
    return {"metric": "conformer-steps/sec (fwd+bwd) + MAE(E,F) vs CPU reference", "value": 37870.123456789, "unit": "conformer-steps/s", "n_gpus": 1, "steps": 10, "warmup": 3,
            "ms_per_step": 54.0812345, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 (" + "y" * 150 + ")", "data": "synthetic",
            "config": {"workload": "w" * 900, "conformers_per_gpu": 2048, "atoms_per_step_per_gpu": 85576.25, "edges_last_step": 1737976, "parallelism": "dp1"},
            "final_loss": 1.7, "gemm_engine": {"what": "z" * 2000},
            "roofline": {"kernel": "k_msgf_rev<true, 2>", "bound": "hbm", "achieved": 452.1, "peak": 8000.0, "unit": "GB/s", "frac": 0.0565, "traffic": 6.4e9,
                         "traffic_source": "profiles/r04_pmc_traffic.json", "traffic_GBps": 3650.0, "algorithmic_bytes_per_launch": 7.845e8, "avg_launch_ms": 1.7357,
                         "launches_per_step": 6, "device_ms_per_step_all_kernels": 53.2, "step": step},
            "cpu_baseline": {"value": 19.4, "unit": "conformer-steps/s", "cores": 16, "host_cpus": 256, "kind": "port", "sample": "s" * 600},
            "mae_vs_cpu_reference": {"mae_energy": 1e-6, "mae_forces": 2e-7, "max_rel_energy": 1e-7, "max_rel_forces": 3e-6, "rel_loss": 1e-7, "max_rel_grad": 2e-6,
                                     "mean_abs_energy_ref": 20.1, "mean_abs_forces_ref": 3.7},
            "sibling_config": {"kernel_ms_per_step": big}, "hamiltonian": {"batch16": big}, "gemnet_oc": {"batch16": big}, "escn": big, "equiformer_v2": big,
            "reference_batch_size_32": {"conformers_per_step": 32, "value": 8506.0, "unit": "conformer-steps/s", "ms_per_step": 3.76, "what": "q" * 500},
            "kernel_ms_per_step": big}


def test_compact_record_is_small_and_round_trips():
    import bench
    full = _full_record()
    assert len(json.dumps(full)) > 20000
    rec = bench.compact_record(full, "gpurun_out/bench_full.json")
    line = json.dumps(rec)
    assert len(line) < bench.COMPACT_LIMIT < 8192
    assert "\n" not in line
    back = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in back, key
    assert back["config"]["workload"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in back["roofline"], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in back["cpu_baseline"], key
    assert abs(back["roofline"]["frac"] - back["roofline"]["achieved"] / back["roofline"]["peak"]) < 1e-3
    assert abs(back["value"] - full["value"]) / full["value"] < 1e-5


def test_compact_record_without_optional_parts():
    import bench
    full = {k: v for k, v in _full_record().items() if k not in ("roofline", "cpu_baseline", "mae_vs_cpu_reference", "kernel_ms_per_step")}
    rec = bench.compact_record(full)
    assert "roofline" not in rec and len(json.dumps(rec)) < bench.COMPACT_LIMIT


def test_stale_pmc_file_is_refused(tmp_path, monkeypatch):
    """roofline.traffic comes from a committed PMC summary only if that summary names a kernel of THIS build at THIS batch size."""
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "r09_pmc_traffic.json").write_text(json.dumps({"batch": 2048, "kernels": {"k_not_a_kernel_of_this_build<true, 2>": {"fetch_kb_per_launch": 1.0,
                                                                                                                                "write_kb_per_launch": 1.0}}}))
    (tmp_path / "nabladft_amd").mkdir()
    (tmp_path / "nabladft_amd" / "libnablaq.so").write_bytes(b"...k_msgf_rev...")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    val, why = bench.pmc_traffic_bytes("k_not_a_kernel", 2048)
    assert val is None and "not in this build" in why
    val, why = bench.pmc_traffic_bytes("k_msgf_rev<true", 2048)
    assert val is None and "does not hold" in why
    val, why = bench.pmc_traffic_bytes("k_not_a_kernel", 1024)
    assert val is None and "batch" in why
    (prof / "r09_pmc_traffic.json").write_text(json.dumps({"batch": 2048, "kernels": {"k_msgf_rev<true, 2>": {"fetch_kb_per_launch": 1000.0, "write_kb_per_launch": 500.0}}}))
    val, why = bench.pmc_traffic_bytes("k_msgf_rev<true", 2048)
    assert val == 1024.0 * 2500.0 and "r09_pmc_traffic.json" in why


def test_compact_record_carries_the_round_5_fields():
    """the rocprofv3 launch average next to the HIP-event figure, the sustained leg, config 2 as written as ONE number marked unpinned,
    and the collective object with the rank count RCCL itself reports."""
    import bench
    full = _full_record()
    full["roofline"].update({"rocprof_avg_launch_us": 1206.9, "rocprof_source": "profiles/r05_rocprofv3_kernel_stats_painn_b2048.csv"})
    full["sustained"] = {"value": 39000.0, "unit": "conformer-steps/s", "steps": 160, "seconds": 8.4, "what": "n" * 400}
    full["sibling_config"] = {"workload": "BASELINE.json configs[1] as written: config/painn.yaml -> schnetpack PaiNN" + "v" * 300, "value": 39031.9, "ms_per_step": 52.47,
                              "parity": "unpinned (third-party arithmetic restated)", "kernel_ms_per_step": {f"k{i}": float(i) for i in range(300)}}
    full["config"]["collective"] = {"backend": "nccl", "ranks_seen": 8, "path": "native", "name": "rccl", "allreduce_exposed_ms": 0.41}
    rec = json.loads(json.dumps(bench.compact_record(full, "gpurun_out/bench_full.json")))
    assert rec["roofline"]["rocprof_avg_launch_us"] == 1206.9 and rec["roofline"]["rocprof_source"].startswith("profiles/r05_")
    assert rec["sustained"] == {"value": 39000.0, "unit": "conformer-steps/s", "steps": 160, "seconds": 8.4}
    assert set(rec["sibling_config"]) == {"workload", "value", "ms_per_step", "parity"} and rec["sibling_config"]["parity"] == "unpinned"
    assert rec["config"]["collective"]["ranks_seen"] == 8 and rec["config"]["collective"]["allreduce_exposed_ms"] == 0.41
    assert len(json.dumps(rec)) < bench.COMPACT_LIMIT
