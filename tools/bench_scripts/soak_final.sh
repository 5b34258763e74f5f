# Soak of the committed library on one GPU box (gpurun from the repository root): every fuzzer with fresh seeds, each step under its own
# timeout, joined with &&; appends to gpurun_out/soak_final.txt.
set -e
O=gpurun_out/soak_final.txt
mkdir -p gpurun_out
echo "== soak of $(git rev-parse --short HEAD 2>/dev/null || echo 'the snapshot') ==" >> $O
timeout -k 10 400 python tests/fuzz_ordered.py 15000 5000 >> $O 2>&1 && echo "ordered done" && \
timeout -k 10 300 python tests/fuzz_diffusion.py 14100 2500 >> $O 2>&1 && echo "diffusion done" && \
timeout -k 10 300 python tests/fuzz_diffusion.py 14200 2500 9,10,12,13,16 >> $O 2>&1 && echo "diffusion small palettes done" && \
timeout -k 10 400 python tests/fuzz_diffusion.py 14300 3000 17,20,31,32,33,64,65,100,128,200,255,256,257,300,700,1024 >> $O 2>&1 && echo "diffusion 17..256 colours (hierarchical nearest table) done" && \
timeout -k 10 300 python tests/fuzz_kmeans.py 15000 1500 hist >> $O 2>&1 && echo "kmeans hist done" && \
timeout -k 10 300 python tests/fuzz_kmeans_fused.py 700 >> $O 2>&1 && echo "kmeans fused done" && \
timeout -k 10 300 python tests/fuzz_distinct.py 19 300 >> $O 2>&1 && echo "distinct done" && \
grep -v amdgpu.ids $O | tail -12
