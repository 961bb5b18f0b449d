"""Host-side logic of bench.py that needs no GPU: the round-over-round comparison block."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _bench():
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        import bench
    finally:
        sys.argv = argv
    return bench


def test_previous_round_values_come_from_the_committed_records():
    bench = _bench()
    rnd, prev = bench.previous_round_values()
    recs = sorted(f for f in os.listdir(ROOT) if f.startswith("BENCH_r") and f.endswith(".json"))
    assert recs and rnd == recs[-1][6:9]
    with open(os.path.join(ROOT, recs[-1])) as f:
        rec = json.load(f)
    assert prev["value"] == (float(rec["parsed"]["value"]), recs[-1] + ":parsed.value")
    for k, (v, src) in prev.items():
        assert v > 0 and (src.startswith("BENCH_r") or src.startswith("profiles/")), (k, v, src)
    assert any(k.startswith("other_robots.") for k in prev)


def test_vs_previous_round_flags_what_fell_and_only_that():
    bench = _bench()
    rnd, prev = bench.previous_round_values()
    # a line that repeats the previous round's numbers exactly, except one robot 5 % down and the host-bound loop 6 % down
    robots = {k.split(".", 1)[1]: {"env_steps_per_s": v} for k, (v, _) in prev.items() if k.startswith("other_robots.")}
    victim = sorted(robots)[0]
    robots[victim]["env_steps_per_s"] *= 0.95
    robots["process"] = "fresh child process of bench.py"          # (a string entry rides in the dict)
    line = {"value": prev["value"][0], "other_robots": robots, "preconditioned": {"value": prev["value"][0] * 1.001}}
    if "api_step_loop" in prev:
        line["api_step_loop_env_steps_per_s"] = {"value": prev["api_step_loop"][0] * 0.94}
    out = bench.vs_previous_round(line)
    assert out["previous"] == rnd
    assert out["regressions"] == ["other_robots." + victim]         # the api loop's bar is -10 %: host-bound
    row = out["values"]["other_robots." + victim]
    assert abs(row["change"] + 0.05) < 1e-3 and row["previous"] > row["now"]
    if "api_step_loop" in prev:
        assert out["values"]["api_step_loop"]["threshold"] == bench.HOST_BOUND_THRESHOLD
    if rnd <= "r04":   # rounds 2-4 reported the preconditioned repetition as `value`: compared like for like
        assert out["values"]["value"].get("like_for_like") is False and "preconditioned" in out["values"]


def test_comparable_values_ignore_what_is_not_a_rate():
    bench = _bench()
    vals = bench._comparable_values({"value": 1.0, "other_robots": {"a": {"env_steps_per_s": 2.0}, "process": "x", "b": {"error": "y"}},
                                     "closed_loop_policy_env_steps_per_s": {"error": "z"},
                                     "closed_loop_policy_wider_env_steps_per_s": {"hidden_128": 3.0, "hidden_256": "err"}})
    assert vals == {"value": 1.0, "other_robots.a": 2.0, "closed_loop_policy.hidden_128": 3.0}
