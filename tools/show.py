import json, sys
tag = sys.argv[1]
for n in [f'{tag}_base', f'{tag}_m16'] + [f'{tag}_b{m}' for m in (1, 2, 4, 8, 16)] + [f'{tag}_m16_b{m}' for m in (1, 2, 4, 8, 16)]:
    try:
        d = json.load(open(f'gpurun_out/{n}.json'))
        print(n, 'kern_ms %.3f' % d['roofline']['kernel_ms_avg'], 'iter %.3f' % d['ms_per_step'],
              d['buckets'] if n.endswith('_base') else '')
    except Exception as e:
        print(n, 'ERR', e)
