"""Loops of one kernel in a device assembly listing with their static instruction counts:  python scratch/isa_loops.py <listing.s> <mangled-name-prefix> [max_len]"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]; maxlen = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
start = [i for i, l in enumerate(lines) if l.startswith(name)][0]
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
body = lines[start:end]
print(len(body), 'lines', sum(1 for l in body if l.strip().startswith('v_')), 'valu static')
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
for i, l in enumerate(body):
    m = re.search(r's_c?branch\S*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] < maxlen:
        a = labels[m.group(1)]; seg = body[a:i + 1]
        print(m.group(1), a, i, 'valu', sum(1 for l in seg if l.strip().startswith('v_')), 'salu', sum(1 for l in seg if l.strip().startswith('s_')), 'lds', sum(1 for l in seg if l.strip().startswith('ds_')),
              'vmem', sum(1 for l in seg if l.strip().startswith(('global_', 'scratch_', 'buffer_', 'flat_'))))
for l in lines[end:end + 60]:
    if any(k in l for k in ('NumVgprs', 'Occupancy', 'ScratchSize', 'NumSgprs')): print(l.strip())
