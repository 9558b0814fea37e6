set -e
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "gptq or gemm or linear" 2>&1 | tail -3
for nw in 4 8; do for s in 1 2 4 8; do echo "NW=$nw S=$s"; FERRUM_HIP_W4_LDSA_NW=$nw FERRUM_HIP_W4_LDSA_S=$s timeout -k 10 300 python tools/exp_dense.py 2>&1 | grep "m=32"; done; done
