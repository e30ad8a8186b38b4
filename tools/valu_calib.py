"""Integer-VALU peak calibration: every mode of apds_dev_valu_peak at 1/2/4/8 waves per SIMD.
Prints wall-clock lane-ops/s, lanes per clock per CU at the s_memtime-derived rate, and cycles per wave-instruction per SIMD."""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("cubesat-apds_amd")
L = pkg._lib.lib()
check = pkg._lib.check
print("mode  instruction                        waves/SIMD   T lane-ops/s   cyc/wave-inst/SIMD   lanes/clk/CU (=4*64*k/cyc)   implied clock GHz")
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for mode in range(first, L.apds_dev_valu_peak_modes()):
    for w in (1, 2, 3, 4, 5, 8):
        rate, cyc, name = C.c_double(0), C.c_double(0), C.c_char_p()
        check(L.apds_dev_valu_peak(mode, w, C.byref(rate), C.byref(cyc), C.byref(name)))
        k = 2 if b"pk_" in name.value else 1          # lane-ops per instruction per lane
        lanes_clk_cu = 4 * 64 * k / cyc.value
        ghz = rate.value / (256 * lanes_clk_cu) / 1e9
        print(f"{mode:3d}   {name.value.decode():34s} {w:6d}      {rate.value / 1e12:9.3f}      {cyc.value:10.3f}          {lanes_clk_cu:10.1f}                 {ghz:6.3f}", flush=True)
