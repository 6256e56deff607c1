"""bench.py's record-keeping (no GPU): the flat `summary` object the driver's record keeps, and the
tie between `roofline.traffic` and the kernel sources the PMC passes ran on."""
import json
import os

import bench
from conftest import ROOT


def test_summary_is_flat_short_and_last():
    shard = {"ms_per_iteration": 0.5123456, "env_steps_per_s": 1.23456789e9,
             "roofline": {"kernel_ms": 0.50123, "frac": 0.014321,
                          "issue_bound": {"frac": 0.93123, "measured_over_priced": 1.3061,
                                          "measured_over_priced_full_chip": 1.1683}}}
    aux = {k: shard for k in ("shard_n3_256_directions", "shard_n6_256_directions",
                              "ars_2048_directions_one_gpu", "ars_2048_directions_one_gpu_n6")}
    aux.update(step_only={"envs_8192": {"us_per_launch": 3.2123, "env_steps_per_s": 2.5e9},
                          "envs_4194304": {"hbm_frac": 0.6912}},
               rollout_saturated={"hbm_frac": 0.6171}, rollout_saturated_65536={"hbm_frac": 0.7172},
               rollout_saturated_n6={"hbm_frac": 0.3121, "env_steps_per_s": 2.2312e10},
               host_cost_at_8_ranks={k: {"host_us_per_iteration": 170.65, "host_bound": False}
                                     for k in ("n3_4096_directions", "n6_2048_directions", "n6_4096_directions")},
               next_rows={"twin_step_envs_4194304": {"hbm_frac": 0.683},
                          "estimator_objective": {"us_per_evaluation": 127.04},
                          "ars_v1_iteration": {"ms_per_iteration": 0.2301},
                          "ars_top_b_64_iteration": {"ms_per_iteration": 0.2931}},
               collective_one_rank={"collective_us": 16.12, "ms_per_iteration": 0.27231},
               collective_one_rank_direct={"collective_us": 9.12, "ms_per_iteration": 0.26831},
               single_env={"gym_step_us": 25.01, "env1_step_us": 12.02, "rlglue_env_step_us": 13.03,
                           "reference_cpu_us_per_step": 111.0})
    line = {"ms_per_step": 0.260212, "value": 3.9351e9,
            "roofline": {"kernel_ms": 0.25231, "frac": 0.03251, "issue_bound_frac": 0.951,
                         "instructions_per_step": 124, "traffic_stale": False},
            "cpu_baseline": {"value": 5.22e7, "cores": 16}}
    sm = bench.summary(line, aux)
    assert all(isinstance(v, (int, float, bool)) for v in sm.values())      # scalars only
    assert len(json.dumps(sm)) < 1900          # the driver's record keeps the last 2000 characters of stdout
    for key in ("n6_sh256_ms", "n6_sh256_sps", "n6_sh256_hbm_frac", "n6_sh256_issue_frac", "n3_sh256_ms",
                "n3_2048_1gpu_ms", "n6_2048_1gpu_over_priced", "step8192_us", "step4m_hbm_frac",
                "sat262144_hbm_frac", "sat65536_hbm_frac", "coll1_us", "gym_step_us", "rlglue_step_us",
                "n6_2048_1gpu_over_priced_full_chip", "sat_n6_hbm_frac", "host_us_n6_4096", "host_bound_n6_4096"):
        assert key in sm
    assert sm["n3_ms"] == 0.2602 and "n_gpus" not in sm                   # main leg: n = 3 unless the config says otherwise
    multi = dict(line, n_gpus=8, config={"segments": 6})
    sm8 = bench.summary(multi, {"collective_us": 21.5, "strong_2048_directions": {"ms_per_iteration": 0.51,
                                                                               "env_steps_per_s": 8.0e9}})
    assert sm8["n6_ms"] == 0.2602 and sm8["n_gpus"] == 8 and sm8["collective_us"] == 21.5 and sm8["strong2048_ms"] == 0.51
    # a leg that failed ({"error": ...}) drops its keys instead of raising
    aux["shard_n6_256_directions"] = {"error": "boom"}
    assert "n6_sh256_ms" not in bench.summary(line, aux)
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.index('line["summary"] = summary(line, aux)') > src.index('line["cpu_baseline"]')


def test_traffic_is_tied_to_the_kernel_sources(tmp_path, monkeypatch):
    h = bench.kernel_source_hash()
    assert len(h) == 64 and h == bench.kernel_source_hash()
    prof = tmp_path / "profiles"
    prof.mkdir()
    kern = bench.rollout_kernel_name(3)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: h)
    (prof / "r03_pmc_traffic.json").write_text(json.dumps({"source_sha256": h, kern: {"traffic_bytes": 7}}))
    t = bench.pmc_traffic(kern, 3, 512, 1000)
    assert t["traffic_bytes"] == 7 and t["stale"] is False
    (prof / "r03_pmc_traffic.json").write_text(json.dumps({"source_sha256": "0" * 64, kern: {"traffic_bytes": 7}}))
    assert bench.pmc_traffic(kern, 3, 512, 1000)["stale"] is True
    (prof / "r03_pmc_traffic.json").write_text(json.dumps({kern: {"traffic_bytes": 7}}))   # no hash recorded
    assert bench.pmc_traffic(kern, 3, 512, 1000)["stale"] is True
    assert bench.pmc_traffic(kern, 3, 256, 1000) is None           # another workload than the measured one


def test_source_hash_ignores_comments_but_not_code(tmp_path, monkeypatch):
    strip = bench._strip_c_comments
    a = 'int a = 1; // note "x"\n/* block\n comment */ const char *s = "// kept /* kept */";  char c = \'"\';\n'
    b = 'int a = 1;\nconst char *s = "// kept /* kept */"; char c = \'"\'; // another note\n'
    assert strip(a) == strip(b)
    assert strip(a) != strip(a.replace("a = 1", "a = 2"))
    assert '"// kept /* kept */"' in strip(a)
