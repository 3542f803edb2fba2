"""Scratch-memory operations inside the solve loop of a step kernel's assembly listing (tools/dbg/loop_spills.sh)."""
import re
import sys

body = open(sys.argv[1]).read().split('\n')
labels = {}
for j, x in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', x)
    if m:
        labels[m.group(1)] = j
loops = []
for j, x in enumerate(body):
    m = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)', x)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < j:
            loops.append((labels[t], j))
# the register-resident solve is recognised by its packed-fp32 friction steps, the streamed-row solve by its reductions
key = 'v_pk_fma' if any('v_pk_fma' in x for x in body) else 'row_bcast:31'
total = sum(1 for x in body if key in x)
best = None
for a, b in loops:
    if sum(1 for x in body[a:b] if key in x) >= total * 0.6 and (best is None or (b - a) < (best[1] - best[0])):
        best = (a, b)
print('scratch operations in the whole kernel:', sum(1 for x in body if 'scratch_' in x))
if best:
    a, b = best
    inside = [j for j in range(a, b) if 'scratch_' in body[j]]
    print('iteration loop: lines %d..%d (%d instructions), scratch operations inside: %d' % (a, b, b - a, len(inside)))
    for j in inside:
        print('   ', j, body[j].strip()[:90])
