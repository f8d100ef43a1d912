"""The HIP runtime's stream -> hardware-queue rule that geotrax_amd.engine.StreamPlan is built on, re-measured on the box the
tests run on: marker kernels on streams created in the plan's order, under rocprofv3, grouped by the Queue_Id column
(tools/stream_map_probe.py). A runtime that deals streams differently fails here instead of silently costing 15-20 % of the
frame rate (profiles/r04_stream_plan.txt)."""
import os
import re
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def _measured_queues(scenario: str, tmp_path: Path) -> dict:
    out = tmp_path / "trace"
    env = dict(os.environ, TMPDIR=str(tmp_path))
    cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", str(out), "--", sys.executable, str(ROOT / "tools" / "stream_map_probe.py"), scenario]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=tmp_path, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-2000:]
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "stream_map_probe.py"), "--read", str(out)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr[-2000:]
    got = {int(k): q for k, q in re.findall(r"s(\d+)->q([\d,]+)", r.stdout)}
    assert got and all("," not in q for q in got.values()), r.stdout      # every stream stays on one queue
    return got


@pytest.mark.gpu
@pytest.mark.parametrize("n_dets,n_stab", [(2, 4), (1, 2)])
def test_the_planned_order_lands_on_the_hardware_queues_the_rule_predicts(tmp_path, n_dets, n_stab):
    if shutil.which("rocprofv3") is None:
        pytest.skip("rocprofv3 not on PATH")
    sys.path.insert(0, str(ROOT / "geo-trax_amd"))
    from geotrax_amd.engine import plan_stream_order, queue_of_streams

    order = plan_stream_order(n_dets, n_stab)
    predicted = queue_of_streams(order)
    steps, streams = [], []                                     # stream k of the probe = the k-th non-null token of the order
    for tok, q in zip(order, predicted):
        if tok == "n":
            steps.append("n0")
        else:
            steps.append(f"c{len(streams)}")
            streams.append((tok, q))
    steps += [f"t{k}" for k in range(len(streams))]
    got = _measured_queues(",".join(steps), tmp_path)
    assert sorted(got) == list(range(len(streams)))
    # same partition: two streams share a measured queue exactly when the rule says they share one
    for a in range(len(streams)):
        for b in range(a + 1, len(streams)):
            assert (got[a] == got[b]) == (streams[a][1] == streams[b][1]), (order, predicted, got)
    # and what the plan is for: a detector's queue carries nothing that ever runs a kernel besides the detector
    det_q = {got[k] for k, (tok, _) in enumerate(streams) if tok == "d"}
    assert len(det_q) == n_dets
    assert all(tok in ("d", "x") or got[k] not in det_q for k, (tok, _) in enumerate(streams)), (order, got)


@pytest.mark.gpu
def test_the_plan_checks_itself_at_run_time(caplog):
    """StreamPlan.verify(): the detector streams of the plan really run at the same time (gtx_streams_overlap: one idle wave
    spinning on each), a stream next to itself does not; a plan whose detector streams were forced onto one hardware queue
    (GTX_ENGINE_ORDER) finds a replacement that overlaps and says so -- in a child process, the plan is per process."""
    sys.path[:0] = [str(ROOT / "geo-trax_amd")]
    from geotrax_amd.engine import StreamPlan

    plan = StreamPlan.get(0, 2, 4)
    assert plan.verified is True
    d = [e[1] for e in plan.ctxs if e[0] == "d"]
    one, two = plan.overlap(d[0], d[1], spin_us=150.0)
    assert 0.1 < one < 0.6 and two < 1.5 * one, (one, two)          # 150 us of spinning, side by side
    one, two = plan.overlap(d[0], d[1])                              # verify()'s own spin: 1 ms (round 6: above a loaded host's jitter)
    assert 0.8 < one < 1.6 and two < 1.5 * one, (one, two)
    one, two = plan.overlap(d[0], d[0])
    assert two > 1.7 * one, (one, two)                               # the same stream: one after the other

    code = ("import sys, logging\n"
            f"sys.path[:0] = [r'{ROOT / 'geo-trax_amd'}']\n"
            "logging.basicConfig(level=logging.WARNING)\n"
            "from geotrax_amd.engine import StreamPlan, queue_of_streams\n"
            "order = ['d', 'x', 'x', 'x', 'x', 'x', 'x', 'd']\n"    # by the rule streams 5-8 join queues 3, 2, 1, 0: the second detector lands on the first's queue
            "assert queue_of_streams(order)[0] == queue_of_streams(order)[-1], queue_of_streams(order)\n"
            "import os; os.environ['GTX_ENGINE_ORDER'] = ','.join(order)\n"
            "plan = StreamPlan.get(0, 2, 0)\n"
            "d = [e[1] for e in plan.ctxs if e[0] == 'd']\n"
            "one, two = plan.overlap(d[0], d[1])\n"
            "print('VERIFIED', plan.verified, two < 1.5 * one)\n")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "VERIFIED False True" in p.stdout, p.stdout + p.stderr   # the plan was wrong, noticed it, and repaired itself
    assert "replacement stream" in p.stderr
