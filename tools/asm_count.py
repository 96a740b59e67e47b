"""Static instruction counts of one kernel in an assembly listing (make -C iv_interpolation_amd/csrc asm -> build/ivs_api.s).
    python tools/asm_count.py <mangled-name-substring> [listing ...]"""
import re, sys
pat = sys.argv[1]
for path in sys.argv[2:] or ["build/ivs_api.s"]:
    t = open(path).read()
    for m in re.finditer(r"^(\S*%s\S*): +; @" % re.escape(pat), t, re.M):
        i = m.end(); j = t.index(".Lfunc_end", i)
        ins = [l.split()[0] for l in t[i:j].split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        n = lambda p: sum(1 for x in ins if x.startswith(p))
        print(f"{path} {m.group(1)[-48:]}: total {len(ins)} valu {n('v_')} rcp {n('v_rcp')} div {n('v_div_')} cndmask {n('v_cndmask')} "
              f"ds {n('ds_')} vmem {n('global_') + n('buffer_')} scratch {n('scratch_')} salu {n('s_')} waitcnt {n('s_waitcnt')}")
