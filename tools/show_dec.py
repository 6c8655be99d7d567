import json, sys
d = json.loads(sys.stdin.read())
print(d['metric'], d['config']['pixels_per_gpu'], 'ms', d['ms_per_step'], 'stages', d['stages'], 'host_io', d['host_io_ms_per_step'], 'roof', d['roofline']['frac'])
