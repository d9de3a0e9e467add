#!/usr/bin/env python3
"""What building a code's Simulator (graph tables, encoder, 64 pooled messages, decoder handle) and its first calls cost:
the sweep scheduler re-builds simulators when a rank moves between codes (profiles/r06_config5_expected_scaling.txt)."""
import sys, time
sys.path.insert(0, '.')
import torch
import ldpc_toolbox_amd as lt
torch.cuda.init()
for code in ["R1_4", "R1_2", "R3_5", "R9_10", "R1_2", "R1_4"]:
    a = lt.code_alist("dvbs2:" + code)
    t0 = time.perf_counter(); s = lt.Simulator(a, "Minsumf32", "", device=0, pool_size=64, pool_seed=8); t1 = time.perf_counter()
    c = s.run(5.0, 1, 0, 4096, 50); t2 = time.perf_counter()
    c = s.run(5.0, 1, 4096, 4096, 50); t3 = time.perf_counter()
    s.close(); t4 = time.perf_counter()
    e0 = time.perf_counter(); enc = lt.Encoder(a); e1 = time.perf_counter()
    d0 = time.perf_counter(); d = lt.LdpcDecoder(a, "Minsumf32"); d1 = time.perf_counter(); d.close()
    print(f"dvbs2:{code}: Simulator ctor {t1-t0:.2f} s (Encoder alone {e1-e0:.2f} s, LdpcDecoder alone {d1-d0:.2f} s), first run {t2-t1:.2f} s, second run {t3-t2:.3f} s, close {t4-t3:.2f} s", flush=True)
