for f in "-DKVQ_TAIL_TRACE"; do
touch kvq-challenge-cvpr-ntire2024_amd/csrc/tail16.hip
KVQ_EXTRA_HIPCC_FLAGS="$f" timeout 600 python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
echo "== $f"; timeout 100 python tools/tail_trace.py 12544 384 2>&1 | grep -E "kernel|item|proj|stores|MLP|lifetime|waits"
done
