set -e
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r04k_gpu_tests.log 2>&1 || { tail -30 gpurun_out/r04k_gpu_tests.log; exit 1; }
tail -2 gpurun_out/r04k_gpu_tests.log
python tests/soak.py --bam --cases 80 --seed 521 > gpurun_out/r04k_soak_bam.log 2>&1 || { tail -20 gpurun_out/r04k_soak_bam.log; exit 1; }
tail -1 gpurun_out/r04k_soak_bam.log
python tests/soak.py --bam-rp --cases 40 --seed 522 > gpurun_out/r04k_soak_bam_rp.log 2>&1 || { tail -20 gpurun_out/r04k_soak_bam_rp.log; exit 1; }
tail -1 gpurun_out/r04k_soak_bam_rp.log
bash tools/evidence4.sh r04k
