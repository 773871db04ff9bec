python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4
