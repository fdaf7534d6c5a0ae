"""The profile post-processing tools on synthetic rocprofv3 output (no GPU, no profiler needed)."""
import csv
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write(path, header, rows):
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(header)
        w.writerows(rows)


def _trace_rows():
    """Three optimizer steps: [gemm 100 us, gelu 20 us, ce 10 us] each, back to back."""
    rows, t, d = [], 1_000_000, 1
    for _ in range(4):  # four cross-entropy marks delimit three whole steps
        for name, dur in (("Cijk_gemm_kernel", 100_000), ("void pda::(anonymous namespace)::gelu_fwd_kernel<unsigned short>(a)", 20_000),
                          ("void pda::(anonymous namespace)::ce_kernel<unsigned short, 16, false>(b)", 10_000)):
            rows.append([d, name, t, t + dur, 1, 0])
            t += dur
            d += 1
    return rows


def test_kernel_avg_reports_per_step_time_and_launches(tmp_path):
    trace = tmp_path / "x_kernel_trace.csv"
    _write(trace, ["Dispatch_Id", "Kernel_Name", "Start_Timestamp", "End_Timestamp", "Queue_Id", "Stream_Id"], _trace_rows())
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_avg.py"), str(trace), "3"],
                         check=True, capture_output=True, text=True).stdout
    lines = out.strip().splitlines()
    assert lines[0].startswith("wall 0.13 ms per step")
    gemm = next(l for l in lines if "Cijk_gemm_kernel" in l)
    assert "0.100 ms/step" in gemm and "1.0 x" in gemm and "100.0 us" in gemm
    # a name filter keeps only the matching kernels
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_avg.py"), str(trace), "3", "gelu"],
                         check=True, capture_output=True, text=True).stdout
    assert "gelu_fwd_kernel" in out and "Cijk" not in out


def test_kernel_clock_divides_active_cycles_by_eight_dies_and_duration(tmp_path):
    counters = tmp_path / "x_counter_collection.csv"
    trace = tmp_path / "x_kernel_trace.csv"
    # a 400 us kernel at 2.0 GHz: GRBM_GUI_ACTIVE (summed over 8 dies) = 8 * 2.0e9 * 400e-6
    _write(counters, ["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"],
           [[1, "void pda::(anonymous namespace)::wgrad_kernel4<0>(x)", "GRBM_GUI_ACTIVE", 8 * 2.0e9 * 400e-6, 0, 1],
            [2, "tiny_kernel", "GRBM_GUI_ACTIVE", 1000.0, 0, 1]])
    _write(trace, ["Dispatch_Id", "Kernel_Name", "Start_Timestamp", "End_Timestamp"],
           [[1, "void pda::(anonymous namespace)::wgrad_kernel4<0>(x)", 1000, 401_000], [2, "tiny_kernel", 500_000, 505_000]])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_clock.py"), str(counters), str(trace)],
                         check=True, capture_output=True, text=True).stdout
    assert "2.00 GHz" in out and "wgrad_kernel4" in out
    assert "tiny_kernel" not in out  # dispatches under 20 us read high and are not listed


def test_analyze_traces_draws_the_notebook_figures(tmp_path):
    """analyze_traces.py --plots: temporal-breakdown pies and the overlap chart from per-rank traces
    (kernel rows on the ranges' clock, collectives recognised by name)."""
    import json

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def trace(path, with_comm):
        ev = [{"ph": "X", "cat": "user_annotation", "name": "ProfilerStep#4", "pid": 0, "tid": 0, "ts": 0.0, "dur": 1000.0}]
        for i in range(8):
            ev.append({"ph": "X", "cat": "kernel", "name": "gemm", "pid": 0, "tid": 100, "ts": 100.0 * i, "dur": 80.0})
        if with_comm:
            ev.append({"ph": "X", "cat": "kernel", "name": "ncclDevKernel_AllReduce", "pid": 0, "tid": 9000, "ts": 150.0, "dur": 100.0})
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump({"traceEvents": ev}, f)

    trace(str(tmp_path / "traces" / "baseline" / "trace.json"), False)
    trace(str(tmp_path / "traces" / "ddp" / "rank0_trace.json"), True)
    trace(str(tmp_path / "traces" / "ddp" / "rank1_trace.json"), True)
    out = subprocess.run([sys.executable, os.path.join(root, "analyze_traces.py"), "--traces", str(tmp_path / "traces"),
                          "--step", "4", "--plots", str(tmp_path / "plots")], capture_output=True, text=True, cwd=root)
    assert out.returncode == 0, out.stderr
    assert "ddp" in out.stdout and "overlap" in out.stdout
    for name in ("temporal_breakdown.png", "comm_comp_overlap.png"):
        p = tmp_path / "plots" / name
        assert p.exists() and p.stat().st_size > 5000, name
    # 100 us of communication (150..250), 30 + 50 of it under the two compute kernels it spans -> 80 % overlap
    row = [ln for ln in out.stdout.splitlines() if ln.startswith("ddp")][0].split()
    assert abs(float(row[-1]) - 80.0) < 0.5, row


def test_trace_notebook_cells_run_on_the_stored_traces(tmp_path, capsys):
    """analyze_traces.ipynb (the notebook form of analyze_traces.py): its code cells run as they stand on the
    per-rank traces kept under profiles/r3/traces — table, figures, kernel differences. IPython's display is
    replaced, since no notebook front end runs here."""
    import json

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    traces = os.path.join(root, "profiles", "r3", "traces")
    if not os.path.isdir(os.path.join(traces, "ddp")):
        pytest.skip("stored traces not present")
    nb = json.load(open(os.path.join(root, "analyze_traces.ipynb")))
    assert nb["nbformat"] == 4
    shown, env = [], {"__name__": "notebook"}
    cwd = os.getcwd()
    os.chdir(root)
    sys.path.insert(0, root)
    try:
        for cell in nb["cells"]:
            if cell["cell_type"] != "code":
                continue
            src = "".join(cell["source"])
            src = src.replace('TRACES = "outputs/traces"', f"TRACES = {traces!r}")
            src = src.replace('"outputs/plots"', repr(str(tmp_path / "plots")))
            src = src.replace("from IPython.display import Image, display",
                              "Image = lambda filename: filename\ndisplay = shown.append")
            env["shown"] = shown
            exec(compile(src, "analyze_traces.ipynb", "exec"), env)
    finally:
        sys.path.remove(root)
        os.chdir(cwd)
    out = capsys.readouterr().out
    assert "fsdp_full_shard" in out and "=== baseline -> ddp" in out
    assert [os.path.basename(p) for p in shown] == ["temporal_breakdown.png", "comm_comp_overlap.png"]
    assert all(os.path.getsize(p) > 10_000 for p in shown)
