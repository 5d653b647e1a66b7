"""Are the kernels of a translation unit unchanged, instruction for instruction, after an edit?  Compares two device assembly files
(hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off --cuda-device-only -S file.hip -o file.s, once per source tree) function
by function with local labels renumbered by order of appearance.  Used at the end of round 4 (no GPU time left) to add the
rotated-row / staged training kernel beside k_train_fb / k_train_adam / k_train_fit without touching their measured code:
    python tools/isa_compare.py /tmp/old/train.s /tmp/new/train.s"""
import re,sys
def funcs(path):
    out={}; cur=None; buf=[]
    for line in open(path):
        m=re.match(r'^(_Z\w+|k_\w+):\s*(;.*)?$',line)
        if m:
            cur=m.group(1); buf=[]; continue
        if cur is not None:
            if line.startswith('.Lfunc_end'):
                out[cur]=buf; cur=None; continue
            l=line.split(';')[0].rstrip()
            if l.strip(): buf.append(l)
    return out
def norm(buf):
    ids={}
    def sub(m):
        k=m.group(0)
        if k not in ids: ids[k]='L%d'%len(ids)
        return ids[k]
    return [re.sub(r'\.LBB\d+_\d+|\.Lpost_getpc\d+',sub,l) for l in buf]
o=funcs(sys.argv[1]); n=funcs(sys.argv[2])
for k in o:
    a=norm(o[k]); b=norm(n.get(k,[]))
    print(k[:70], len(a), len(b), 'IDENTICAL' if a==b else 'differs')
for k in n:
    if k not in o: print('new:',k[:70],len(n[k]))
