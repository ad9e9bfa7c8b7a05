import sys, json
d=json.loads(sys.stdin.read()); st=d['config'].get('plan', {})
print('%-6s %-4s %-5s %8.3f Gblk/s  step %.4f ms  parity %s' % (sys.argv[1], d['config']['alpha_regime'], d['config'].get('tiles', '?')[:5], d['value']/1e9, d['ms_per_step'], d['parity']))
