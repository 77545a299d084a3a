"""Attribution of the native frames of profiles/r02_pmc_async_abort_stderr.txt to libraries WITHOUT re-running the abort: the low
12 bits of a return address survive ASLR and all frames inside one library share one page-aligned load base, so for every candidate
library the bases under which ALL frames of a group are call-return sites are enumerated (objdump -d).
usage: python tools/attribute_frames.py <lib.so> [<lib.so> ...]"""
import subprocess, sys, re, os, pickle
OBJ="/opt/rocm/lib/llvm/bin/llvm-objdump"
def ret_sites(path):
    cache=os.path.join(os.environ.get("TMPDIR","/tmp"),"attr_"+path.replace("/","_")+".pkl")
    if os.path.exists(cache): return pickle.load(open(cache,"rb"))
    p=subprocess.Popen([OBJ,"-d","--no-show-raw-insn",path],stdout=subprocess.PIPE,text=True)
    sites=set(); prev_call=False
    for line in p.stdout:
        m=re.match(r"\s*([0-9a-f]+):\s+(\S+)",line)
        if not m: continue
        addr=int(m.group(1),16); mn=m.group(2)
        if prev_call: sites.add(addr)
        prev_call = mn.startswith("call")
    pickle.dump(sites,open(cache,"wb"))
    return sites
groups={"A":[0x7c7b2a39bc1a,0x7c7b2a397f89,0x7c7b2a398615,0x7c7b2a362635,0x7c7b2a221475,0x7c7b2a26d284,0x7c7b2a2219ea,0x7c7b2a2389b1],
        "B":[0x7c7c68a3d266,0x7c7c68a2e5c0],
        "C":[0x7c7c735262fb],"D":[0x7c7c73886ec0],"H":[0x7c7c73f3750e],"G":[0x7c7c73072ee8]}
libs=sys.argv[1:]
for lib in libs:
    S=ret_sites(lib)
    print(lib,len(S),"return sites")
    for g,ras in groups.items():
        if len(ras)<2:
            # single frame: just count candidates with same low 12 bits (weak)
            n=sum(1 for s in S if (s&0xfff)==(ras[0]&0xfff))
            print("  group",g,"single frame: ",n,"sites share low12")
            continue
        r0=ras[0]; found=[]
        for s in S:
            if (s&0xfff)!=(r0&0xfff): continue
            B=r0-s
            if all((r-B) in S for r in ras): found.append(B)
        print("  group",g,"consistent bases:",[hex(b) for b in found])
