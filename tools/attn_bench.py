import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import io, contextlib
import microbench as mb
with contextlib.redirect_stdout(io.StringIO()):
    for _ in range(20): mb.bench_attn(32, 12, 2305)
for _ in range(3): mb.bench_attn(32, 12, 2305)
mb.bench_attn(16, 16, 3601)
