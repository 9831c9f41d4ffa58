set -e
scripts/profile_bench.sh metric_k64 --steps 192 --warmup 64 --iters-per-launch 64
scripts/profile_bench.sh metric_k1 --steps 96 --warmup 16 --iters-per-launch 1
scripts/profile_bench.sh c2_k64 --config c2 --steps 192 --warmup 64
scripts/profile_bench.sh c3 --config c3 --steps 48 --warmup 8
scripts/profile_bench.sh c4_k1 --config c4 --steps 48 --warmup 8 --iters-per-launch 1
scripts/profile_bench.sh c4_k16 --config c4 --steps 64 --warmup 32
scripts/profile_bench.sh c5_k64 --config c5 --steps 128 --warmup 64
