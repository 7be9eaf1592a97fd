"""stdin -> only bench.py's JSON line (RCCL prints a version banner on stdout before it)."""
import sys
for line in sys.stdin:
    if line.startswith("{") and '"metric"' in line:
        sys.stdout.write(line)
