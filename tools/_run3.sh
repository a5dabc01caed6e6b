for i in 1 2; do
(cd _old && python tools/step1_time.py 2>/dev/null && python tools/step1_time.py --task 2>/dev/null)
python tools/step1_time.py 2>/dev/null && python tools/step1_time.py --task 2>/dev/null
done
timeout 1500 python -m pytest tests/test_gpu_step1.py -x -q 2>&1 | tail -5
