for i in 1 2; do
(cd _old && python tools/step1_time.py --task 2>/dev/null)
python tools/step1_time.py --task 2>/dev/null; python tools/step1_time.py 2>/dev/null
done
python tools/task_world_time.py 2>&1 | grep "step plan" | tail -2
timeout 600 python -m pytest tests/test_gpu_step1.py tests/test_gpu_task.py tests/test_gpu_task_world.py -x -q 2>&1 | tail -2
