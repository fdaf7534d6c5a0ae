"""The profile post-processing tools on synthetic rocprofv3 output (no GPU, no profiler needed)."""
import csv
import os
import subprocess
import sys

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
