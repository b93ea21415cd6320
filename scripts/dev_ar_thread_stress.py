"""dev: the two-thread hipGraph AR test in a loop (tests/test_gpu_threads.py) -- how often does a capture fail / a call fail?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import test_gpu_threads as t
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = 0
t0 = time.time()
for i in range(n):
    try:
        t.test_one_teacher_handle_two_threads_two_streams(True)
    except Exception as e:           # noqa
        bad += 1
        print('iteration', i, 'FAILED:', str(e)[:300])
print('%d iterations, %d failed, %.1f s' % (n, bad, time.time() - t0))
