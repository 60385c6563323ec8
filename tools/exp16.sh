for f in "-DKVQ_TAIL_TRACE" "-DKVQ_TAIL_TRACE -DT16_NOLDS" "-DKVQ_TAIL_TRACE -DKVQ_TAIL_NOGELU"; do
touch kvq-challenge-cvpr-ntire2024_amd/csrc/tail16.hip
KVQ_EXTRA_HIPCC_FLAGS="$f" timeout 600 python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
echo "== $f"; timeout 100 python tools/tail_trace.py 12544 384 2>&1 | grep -E "kernel|MLP|lifetime|waits"
done
