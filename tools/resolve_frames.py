"""Which shared library do the raw return addresses of a crash log belong to?  For every library given on the command line and every
group of addresses below, list the page-aligned load bases under which ALL addresses of the group directly follow a call instruction
of the library's executable segments (vaddr-correct via the program headers).  Used for profiles/README.md (round-2 rocprofv3 abort)."""
import sys,re,subprocess
def segs(lib):
    out=subprocess.check_output(['readelf','-lW',lib]).decode()
    r=[]
    for l in out.splitlines():
        p=l.split()
        if p and p[0]=='LOAD':
            off,va,_,fsz,msz=[int(x,16) for x in p[1:6]]
            flags=''.join(p[6:-1])
            r.append((off,va,fsz,'E' in flags))
    return r
def exec_vaddr_calls(lib):
    data=open(lib,'rb').read()
    s=set()
    for off,va,fsz,ex in segs(lib):
        if not ex: continue
        blob=data[off:off+fsz]
        for m in re.finditer(b'\xe8', blob): s.add(va+m.start()+5)
        for m in re.finditer(b'\xff[\x10-\x17\x50-\x57\x90-\x97\xd0-\xd7]', blob):
            for l in (2,3,6,7): s.add(va+m.start()+l)
    return s
groups={'early':[0x7f8005fd05c0,0x7f8005fdf266],
 'hipA':[0x7f7ffc9f19b1,0x7f7ffc9da9ea,0x7f7ffca26284,0x7f7ffc9da475],'hipB':[0x7f7ffcb1b635,0x7f7ffcb51615,0x7f7ffcb50f89,0x7f7ffcb54c1a],
 'hipAB':[0x7f7ffc9f19b1,0x7f7ffc9da9ea,0x7f7ffca26284,0x7f7ffc9da475,0x7f7ffcb1b635,0x7f7ffcb51615,0x7f7ffcb50f89,0x7f7ffcb54c1a],
 'top':[0x7f8010e28ec0,0x7f80114d950e]}
for lib in sys.argv[1:]:
    cs=exec_vaddr_calls(lib)
    mx=max(cs)
    for g,addrs in groups.items():
        a0=addrs[0]; hits=[]
        for va in cs:
            if (va&0xfff)==(a0&0xfff):
                base=a0-va
                if all((a-base) in cs for a in addrs[1:]): hits.append((hex(base),hex(va)))
        print(lib.split('/')[-1], g, hits[:4], len(hits))
