"""Prints the in-process ceilings (lance_hip_ubench) as one JSON line: LDS random-gather rates for 4/8/16-byte LUT entries,
device copy bandwidth, f32 VALU wave-instruction issue rates.  GPU only."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lance_amd  # noqa: E402

eng = lance_amd.default_engine()
LDS = ("lds4", "lds8", "lds16", "lds8_u16x4", "lds8_stagger16", "lds8_stagger32", "lds8_linear")
res = {k: eng.ubench(k) for k in LDS + ("copy", "valu", "valu_pk")}
cus = 256
res["per_clk_per_cu_at_2.4GHz"] = {k: res[k] / cus / 2.4e9 for k in LDS}
res["guide_ds_read_b64_peak_lane_gathers_per_s"] = 32 * cus * 2.4e9     # MI355X_MICROARCH.md: ds_read_b64 256 B/clk/CU
res["valu_cycles_per_wave_instr_per_simd_at_2.4GHz"] = {k: cus * 4 * 2.4e9 / res[k] for k in ("valu", "valu_pk")}
print(json.dumps(res))
