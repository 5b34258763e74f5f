mkdir -p gpurun_out/r4g
for s in 300 301 302 303 304 305 306 307 308 309 310 311; do
  timeout -k 10 300 python tests/fuzz_diffusion.py $s 250 9,10,12,13,16 >> gpurun_out/r4g/soak_ed.txt 2>&1 || exit 1
done
tail -14 gpurun_out/r4g/soak_ed.txt
