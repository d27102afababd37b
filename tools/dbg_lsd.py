import sys, importlib, ctypes, numpy as np
sys.path.insert(0, '/root/repo'); import __graft_entry__ as ge
hvo = ge.package(); synth = importlib.import_module("hvo_amd.synth")
ctx = hvo.Context()
for kind, seed in (("std", 0x5EED0002), ("std", 0x5EED1001), ("lowtex", 0x5EED0001)):
    g = synth.make_gray(kind, seed)
    ctx.extract_lsd(g); kl, _, _ = ctx.extract_lsd(g)
    out = (ctypes.c_longlong * 8)()
    L = hvo.lib(); L.hvo_debug_lsd_stats.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.hvo_debug_lsd_stats(ctx.h, 0, out)
    s = list(out)
    print(kind, "seeds %d pts %d big %d | grow %.2f ms rect %.2f ms refine %.2f ms total %.2f ms segs %d" % (s[0], s[1], s[2], s[3]/1e5, s[4]/1e5, s[5]/1e5, s[6]/1e5, s[7]))
