"""TEST INFRASTRUCTURE ONLY: the CPU oracle (plain-C restatement, `liboracle.so`) and the real reference
library (`_ref/libggml_ref_*.so`).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this package; the product (ik_llama.cpp_amd/) never does."""
