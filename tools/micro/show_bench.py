import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('value %.4g  ms/step %.4f  roofline frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))
for k in ('pair_stage', 'wider_layers', 'hybrid_head'):
    if k in d:
        print(k, json.dumps(d[k])[:700])
