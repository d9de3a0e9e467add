#!/usr/bin/env python3
"""Register / LDS / scratch use of the kernels in any hipcc-built binary or library (default: the product library),
read from the embedded gfx950 code objects' metadata notes (no GPU needed).
usage: tools/co_resources.py [--lib PATH] [substring of the demangled kernel name ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources as kr
if len(sys.argv) > 2 and sys.argv[1] == "--lib":
    os.environ["LDPC_LIB"] = sys.argv[2]
    del sys.argv[1:3]
kr.main()
