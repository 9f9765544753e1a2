# window rates at the other shapes the configurations name (one box): ViT-L 518x924 (720p / 1080p sources), ViT-L 1036x1036,
# ViT-S 518x686 (480x640 cameras), both modes where they matter; long videos through the whole-video path
set -e
cd $GRAFT_REPO_ROOT
P=${1:-r3}
B="python bench.py --no-end-to-end --no-cpu-baseline --no-vits --no-fp32-mode --no-kernel-timing"
: > gpurun_out/${P}_shapes.jsonl
timeout -k 10 200 $B --height 518 --width 924 --steps 5 --warmup 2 >> gpurun_out/${P}_shapes.jsonl
timeout -k 10 200 $B --height 1036 --width 1036 --steps 3 --warmup 1 >> gpurun_out/${P}_shapes.jsonl
timeout -k 10 200 $B --encoder vits --height 518 --width 686 --steps 10 --warmup 3 >> gpurun_out/${P}_shapes.jsonl
timeout -k 10 200 $B --encoder vits --height 1036 --width 1036 --steps 5 --warmup 2 >> gpurun_out/${P}_shapes.jsonl
timeout -k 10 300 $B --precision fp32 --height 518 --width 924 --steps 3 --warmup 1 >> gpurun_out/${P}_shapes.jsonl
timeout -k 10 300 $B --precision fp32 --encoder vits --steps 10 --warmup 2 >> gpurun_out/${P}_shapes.jsonl
timeout -k 10 300 python tools/long_video.py 300 vitl 720 1280 > gpurun_out/${P}_long_video.txt 2>&1
timeout -k 10 300 python tools/long_video.py 2000 vits 480 640 >> gpurun_out/${P}_long_video.txt 2>&1
