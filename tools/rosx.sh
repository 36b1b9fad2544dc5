for d in "" "RMT_ROS_WAVES=2" "RMT_ROS_LOGTOL=(-18.4)" "RMT_ROS_LOGTOL=(-13.8)"; do echo "== $d"; timeout -k 10 200 python tools/tts_bench.py 256 1024 0.5 SKIP_RK4=1 $d 2>/dev/null | tail -3; done
