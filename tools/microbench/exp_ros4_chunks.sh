set -e
mkdir -p gpurun_out/e7
L=gpurun_out/e7/chunks.txt
: > $L
run() { echo "## $*" >> $L; "$@" >> $L 2>&1; }
for spec in "1024 128 0.1" "1024 64 0.1" "1024 200 0.1" "4096 64 0.05" "4096 16 0.05" "16384 8 0.02"; do
  set -- $spec
  for c in auto 1 2 4 8 16 32 64; do
    nb=$(( ($1 + 255) / 256 ))
    if [ "$c" != "auto" ] && [ "$c" -gt "$nb" ]; then continue; fi
    if [ "$c" = "auto" ]; then unset RMT_N2_ROS4_CHUNKS; else export RMT_N2_ROS4_CHUNKS=$c; fi
    echo "N=$1 E=$2 chunks=$c" >> $L
    python tools/run_one.py ros4 dme_nb $1 $2 $3 256 1 auto >> $L 2>&1
  done
done
unset RMT_N2_ROS4_CHUNKS
grep -v "amdgpu.ids\|^accepted" $L
