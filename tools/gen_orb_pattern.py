#!/usr/bin/env python3
"""Extract the 256x4 learned rBRIEF sampling pattern (a pure data table) from the
reference (src/ORBextractor.cc:148-406, `bit_pattern_31_`) into a flat .inc file.

Only the 1024 integers are emitted (no reference text).  Run in the build container,
where /root/reference exists; the generated files are committed.
"""
import re, sys, hashlib
src = open('/root/reference/src/ORBextractor.cc').read().split('\n')
body = '\n'.join(src[147:406])
body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
body = body[body.index('{') + 1: body.rindex('}')]
vals = [int(t) for t in re.findall(r'-?\d+', body)]
assert len(vals) == 1024, len(vals)
assert vals[:4] == [8, -3, 9, 5] and vals[-4:] == [-1, -6, 0, -11]
txt = ',\n'.join(','.join(str(v) for v in vals[i:i + 32]) for i in range(0, 1024, 32)) + '\n'
for out in sys.argv[1:]:
    open(out, 'w').write(txt)
print('sha256', hashlib.sha256(txt.encode()).hexdigest())
