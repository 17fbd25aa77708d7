"""CPU: the bench line says one thing -- every printed fraction can be recomputed from the printed numbers -- and the
runners' shared loop calls its hooks in the reference loops' order."""
import json
import os
import sys

from conftest import ROOT

sys.path.insert(0, ROOT)


def test_roofline_fields_are_self_consistent(monkeypatch):
    import bench
    pmc = {"kernel_src_sha": bench.kernel_src_sha(), "kernel": "void texir::irt_group_kernel<false, 4, 6>", "rays_per_launch": 1000000,
           "fabric_bytes_per_launch": 3.0e8, "SQ_INSTS_VALU": 5.0e7, "tcp_cache_accesses": 3.0e7, "tcp_clocks": 4.0e7, "kernel_ms_under_pmc": [0.1, 0.1],
           "l2_hit_rate": 0.5, "valu_lane_utilisation": 0.7, "source": "unit test"}
    monkeypatch.setattr(bench, "load_pmc", lambda w, k: (pmc, None))
    monkeypatch.setattr(bench, "load_chain", lambda w, k: (None, "no chain probe in this test yet"))
    r = bench.roofline("c4", "irt_group_kernel<false, 4, 6>", 0.1, 1000000, 1, (1952.0, 54.6, 3.65, 1.0))
    # top level = the memory side, always: frac = achieved / peak, achieved = traffic / time
    assert r["bound"] == "hbm" and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["traffic"] / (r["kernel_ms"] * 1e-3) / 1e9) < 0.1
    # every limit carries its own numerator and denominator, and `binding` names the largest fraction
    fr = {k: v["frac"] for k, v in r["limits"].items() if v}
    for k, v in r["limits"].items():
        if v:
            assert abs(v["frac"] - v["achieved"] / v["peak"]) < 2e-3, k
    assert r["binding"] == max(fr, key=fr.get)
    assert "SELF-CALIBRATED" in r["limits"]["l1"]["note"]
    # with a chain probe of these sources the line carries the dependent-chain bound and the occupancy sweep too
    chain = {"kernel_src_sha": bench.kernel_src_sha(), "kernel": "void texir::irt_group_kernel<false, 4, 6>", "rays_per_launch": 1000000,
             "chain_bound": {"seconds": 6.0e-5}, "reading": {"rate_8_waves_over_1_wave": 4.2},
             "full_occupancy": {"probe_overhead": 0.3, "per_step": {"steps_per_pass": {"node_vector": 12, "node_scalar": 6, "leaf": 3}}},
             "occupancy_sweep": {"spp": 256, "points": [{"waves_per_simd": w, "grays_per_s": g, "cycles_per_step": {"node_vector": 1800.0 * w ** 0.5}} for w, g in ((1, 3.3), (2, 6.0), (4, 10.0), (8, 14.0))]}}
    monkeypatch.setattr(bench, "load_chain", lambda w, k: (chain, None))
    r2 = bench.roofline("c4", "irt_group_kernel<false, 4, 6>", 0.1, 1000000, 1, None)
    assert abs(r2["limits"]["chain"]["frac"] - 0.6) < 1e-3 and r2["occupancy"]["grays_per_s"][-1] == 14.0
    fr2 = {k: v["frac"] for k, v in r2["limits"].items() if v}
    assert r2["binding"] == max(fr2, key=fr2.get)
    assert r["algorithmic"]["bytes_per_ray"] == 1952.0
    json.dumps(r)
    # a profile of other kernel sources is refused, loudly, and the line then carries no measured bound
    monkeypatch.setattr(bench, "load_pmc", lambda w, k: (None, "profiles/pmc_c4.json was taken with other kernel sources"))
    r = bench.roofline("c4", "irt_group_kernel<false, 4, 6>", 0.1, 1000000, 1, None)
    assert r["frac"] is None and "other kernel sources" in r["note"]


def test_committed_profiles_match_the_committed_kernel_sources():
    """profiles/pmc_<workload>.json are only read when they were taken with THESE sources: the committed pair must agree"""
    import bench
    sha = bench.kernel_src_sha()
    for w in ("c4", "c2", "c4_scan", "house", "c1"):
        p = os.path.join(ROOT, "profiles", "pmc_%s.json" % w)
        assert os.path.exists(p), p
        d = json.load(open(p))
        assert d["kernel_src_sha"] == sha, (w, d["kernel_src_sha"], sha)
        assert d["fabric_bytes_per_launch"] > 0 and d["rays_per_launch"] > 0
    # the chain probes (tools/chain_probe.py) of the headline and of its hostile sibling: same rule
    for w in ("c4", "c4_scan"):
        p = os.path.join(ROOT, "profiles", "chain_%s.json" % w)
        assert os.path.exists(p), p
        d = json.load(open(p))
        assert d["kernel_src_sha"] == sha, (w, d["kernel_src_sha"], sha)
        assert 0.0 < d["chain_bound"]["frac_of_shipped_kernel_time"] <= 1.0 and len(d["occupancy_sweep"]["points"]) == 4
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from mat_step_pmc import mat_src_sha
    assert json.load(open(os.path.join(ROOT, "profiles", "pmc_mat_step.json")))["mat_src_sha"] == mat_src_sha()


def test_runner_base_hook_order():
    from texir_code_amd.trainer.base import RunnerBase

    class M:
        def train(self):
            log.append("train")

    log = []
    r = RunnerBase()
    r.model, r.cur_iter = M(), 0
    ended = r.fit([10, 11, 12], 0, 1, lambda b: (log.append("step %d" % b), b)[1], epoch_begin=lambda e: log.append("begin %d" % e),
                  takes=lambda i: i != 1, before_step=lambda e, i: log.append("before %d.%d" % (e, i)),
                  after_step=lambda e, i, out: (log.append("after %d.%d=%d it%d" % (e, i, out, r.cur_iter)), r.cur_iter >= 3)[1],
                  epoch_end=lambda e: log.append("end %d" % e))
    assert ended is True
    assert log == ["begin 0", "train", "before 0.0", "step 10", "after 0.0=10 it1", "train", "before 0.2", "step 12", "after 0.2=12 it2", "end 0",
                   "begin 1", "train", "before 1.0", "step 10", "after 1.0=10 it3"]


def test_fused_adam_groups_are_fixed_at_construction():
    import pytest
    import torch
    from texir_code_amd._lib import TexirError
    from texir_code_amd.optim import FusedAdam
    a, b = torch.nn.Parameter(torch.zeros(4, 4, 3)), torch.nn.Parameter(torch.zeros(3))
    opt = FusedAdam([{"params": [a]}, {"params": [b], "lr": 1e-2}], lr=1e-3)
    assert len(opt.param_groups) == 2 and opt.param_groups[1]["lr"] == 1e-2
    with pytest.raises(TexirError, match="fixed at construction"):
        opt.add_param_group({"params": [torch.nn.Parameter(torch.zeros(2))]})


def test_cpu_baseline_reports_the_cpus_it_may_really_use():
    """the baseline line states affinity, cgroup quota and the thread count derived from them (the round-3 line said "256 cores" on a
    box whose cgroup allowed 16), and the parts-per-texel helper of the footprint figure follows the launcher's rule (kernels.hip irt_plan)"""
    import bench
    h = bench.host_cpus()
    assert 1 <= h["usable"] <= h["affinity"] <= (h["os_cpu_count"] or h["affinity"])
    if h["cgroup_cpus"] is not None:
        assert h["usable"] <= int(h["cgroup_cpus"]) + 1
    assert [bench.irt_plan_parts(n) for n in (2048, 1024, 256, 64, 16, 8, 100)] == [32, 32, 32, 8, 2, 1, 1]


def test_between_steps_hook_sees_the_next_batch_of_the_same_epoch_only():
    """RunnerBase.fit (round 5): between_steps(next_batch, epoch, next_index) runs after a step was launched and before after_step, only when the epoch has a next batch
    this rank takes -- never across an epoch boundary (that is where the reference's loops draw other random numbers) and never after the last step"""
    from texir_code_amd.trainer.base import RunnerBase

    class M:
        def train(self):
            pass

    log = []
    r = RunnerBase()
    r.model, r.cur_iter = M(), 0
    r.fit([10, 11, 12, 13], 0, 1, lambda b: b, takes=lambda i: i != 2,
          after_step=lambda e, i, out: log.append("after %d.%d" % (e, i)), between_steps=lambda nb, e, ni: log.append("prep %d.%d=%d" % (e, ni, nb)),
          epoch_end=lambda e: log.append("end %d" % e))
    # batch 2 is another rank's: no preparation for it; batch 3 is prepared only when it is the NEXT one (after the skipped batch nothing was launched)
    assert log == ["prep 0.1=11", "after 0.0", "after 0.1", "after 0.3", "end 0", "prep 1.1=11", "after 1.0", "after 1.1", "after 1.3", "end 1"]


def test_collate_one_view_equals_default_collate_without_copies():
    import torch
    from texir_code_amd.trainer.train_material import collate_one_view
    item = {"color": torch.rand(6, 4, 4, 3), "mask": torch.rand(6, 4, 4, 1), "cam_to_world": torch.rand(6, 4, 4), "id": "view003", "cam_position": torch.rand(3), "n": 5}
    a, b = collate_one_view([item]), torch.utils.data.default_collate([item])
    assert set(a) == set(b)
    for k in a:
        if torch.is_tensor(b[k]):
            assert a[k].shape == b[k].shape and torch.equal(a[k], b[k]), k
        else:
            assert a[k] == b[k], k
    assert a["color"].data_ptr() == item["color"].data_ptr()                  # a view of the dataset's tensor, not a stacked copy
    two = collate_one_view([item, item])
    assert two["color"].shape == (2, 6, 4, 4, 3)                              # (larger batches fall back to the stock collate)


def _contract_keys():
    return {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline", "full"}


def test_compact_line_fits_the_drivers_capture_and_keeps_the_contract_keys():
    """BENCH_r05.json had parsed = null because the printed line had grown to 20 KB.  The printed line is built by bench.compact_line from the full record;
    canned inputs = committed full records (round 5's 20 KB line, N = 1 driver command; round 6's 2 gloo ranks on one GPU and N = 1 record as the current code writes them)"""
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r05", "bench_driver_command.json")))
    assert len(json.dumps(full)) > 15000
    line = bench.compact_line(full, "gpurun_out/bench_full.json")
    assert "\n" not in line and len(line) < 4096 and len(line) <= bench.LINE_CAP
    o = json.loads(line)
    assert _contract_keys() <= set(o), _contract_keys() - set(o)
    assert set(o["config"]) == {"workload", "parallelism"} and o["config"]["workload"].startswith("c4:")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "data", "scaling", "higher_is_better"):
        assert o[k] == full[k], k
    r = o["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "binding", "binding_frac", "algorithmic_frac", "profile"} <= set(r)
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["traffic"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1.0
    # SURVEY 8(d)'s algorithmic bytes over the live kernel time: recomputable from the line, and > 1 on this path (cache-served; it bounds nothing)
    rays = full["roofline"]["rays_per_launch"]
    assert abs(r["algorithmic_frac"] - r["algorithmic_bytes_per_ray"] * rays / (r["kernel_ms"] * 1e-3) / 8e12) < 5e-3 and r["algorithmic_frac"] > 1.0
    assert r["binding_frac"] == full["roofline"]["limits"][r["binding"]]["frac"] and r["profile"] == "profiles/pmc_c4.json"
    assert set(o["cpu_baseline"]) == {"value", "unit", "cores", "kind", "sample"} and o["cpu_baseline"]["kind"] == "port"
    assert {"ms", "ms_back_to_back", "frac"} <= set(o["material_step"]) and o["material_step"]["ms"] == full["material_step"]["ms"]
    assert set(o["extra_mrays_s"]) == {"c4_scan", "house"} and o["projected_speedup"]["8"] == full["ranks"]["projected"]["8"]["projected_speedup"]
    ph = full["e2e"]["mat"]["phases_s"]
    assert abs(o["e2e_s"]["mat_stage_ms_per_step"] - 1e3 * (ph["stage0"] + ph["stage1"] + ph["stage2"]) / full["e2e"]["mat"]["steps"]) < 1e-3
    assert o["full"] == "gpurun_out/bench_full.json"
    # N > 1 record: same cap, the per-rank block in place of the projection
    full2 = json.load(open(os.path.join(ROOT, "profiles", "r06", "bench_2rank_gloo_one_gpu_full.json")))
    line2 = bench.compact_line(full2, "x.json")
    o2 = json.loads(line2)
    assert len(line2) <= bench.LINE_CAP and (_contract_keys() - {"cpu_baseline"}) <= set(o2)
    assert o2["n_gpus"] == 2 and len(o2["ranks"]["kernel_ms"]) == 2 and o2["ranks"]["assembled_ok"] is True and o2["ranks"]["backend"] == "gloo"
    assert o2["ranks"]["rccl_ranks"] is None and o2["ranks"]["collective_bytes_per_step"] == 12 * 12417 and "projected_speedup" not in o2
    # the printed line of the committed driver-shaped run IS what compact_line makes of its full record
    full6 = json.load(open(os.path.join(ROOT, "profiles", "r06", "bench_driver_command_full.json")))
    printed = open(os.path.join(ROOT, "profiles", "r06", "bench_driver_command.json")).read().strip().splitlines()[-1]
    assert json.loads(bench.compact_line(full6, json.loads(printed)["full"])) == json.loads(printed)
    assert json.loads(printed)["roofline"]["profile_matches_live"] is True and json.loads(printed)["e2e_s"]["mat_stage_ms_per_step"] < 0.80
    # an unexpected long string cannot push the line past the cap: optional blocks go first
    full3 = json.loads(json.dumps(full))
    full3["config"]["workload"] = "w" * 2500
    assert len(bench.compact_line(full3, "x.json")) <= bench.LINE_CAP


def test_write_full_record_goes_to_the_named_file(tmp_path, monkeypatch):
    import bench
    p = tmp_path / "sub" / "full.json"
    monkeypatch.setenv("TEXIR_BENCH_FULL", str(p))
    assert bench.write_full({"a": 1}) == str(p) and json.load(open(p)) == {"a": 1}
