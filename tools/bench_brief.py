"""Walk kernels and step time of every configuration of a bench_details.json, one line each (A/B reading aid)."""
import json
import sys

d = json.load(open(sys.argv[1]))


def brief(name, v):
    k = v.get('kernels', {})
    print('%-18s %7.3f ms  %s' % (name, v.get('ms_per_step') or -1,
                                 '  '.join('%s %.0f' % (n[4:], k[n]['avg_us']) for n in ('nam_first', 'nam_step_sparse', 'nam_step') if n in k)
                                 + ('  gram %.0f null %.0f' % (k['gram']['avg_us'], k['null_local']['avg_us']) if 'gram' in k and 'null_local' in k else '')),
          v.get('error') or '')


brief(d['config']['workload'].split(':')[0], d)
for name, v in d.get('other_configs', {}).items():
    brief(name, v)
