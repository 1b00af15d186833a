"""Numerical emulation of F(4x4,3x3) (fp32) inside the oracle, to see how much parity margin the kernel would leave."""
import sys, importlib, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
import state_bounds as C
import rerevst_oracle as O
pkg = importlib.import_module("rerevst-code_amd")
F32=np.float32
# the kernel's interpolation points 0, +-3/4, +-3/2, inf (rerevst-code_amd/csrc/conv_f43.h; matrices from tools/f43_points.py)
import importlib.util, os
_spec = importlib.util.spec_from_file_location("f43_points", os.path.join(os.path.dirname(os.path.abspath(__file__)), "f43_points.py"))
_fp = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(_fp)
_AT, _G, _BT = _fp.cook_toom([0, .75, -.75, 1.5, -1.5])
BT=_BT.astype(F32); AT=_AT.astype(F32)
G=_G          # float64: the weight transform is evaluated in double and rounded once (pack_f43_k)
orig=O.conv3x3
MODE={"layers":"all"}
def conv_f43(x,w,b=None):
    Bn,H,W,Cin=x.shape; Cout=w.shape[0]
    if Cin<64 or Cout<64 or MODE["layers"]=="none": return orig(x,w,b)
    th,tw=(H+3)//4,(W+3)//4
    xp=np.zeros((Bn,th*4+2,tw*4+2,Cin),F32); xp[:,1:H+1,1:W+1]=x
    # U = G g G^T  [36][Cin][Cout]
    U=np.einsum('ia,ocab,jb->ijco',G,w.astype(np.float64),G,optimize=True).astype(F32).reshape(36,Cin,Cout)
    out=np.zeros((Bn,th*4,tw*4,Cout),F32)
    for bi in range(Bn):
        # patches [th,tw,6,6,Cin]
        s=xp[bi].strides
        pt=np.lib.stride_tricks.as_strided(xp[bi],shape=(th,tw,6,6,Cin),strides=(4*s[0],4*s[1],s[0],s[1],s[2]))
        V=np.einsum('ia,tuabc,jb->ijtuc',BT,pt,BT,optimize=True).astype(F32).reshape(36,th*tw,Cin)
        M=np.matmul(V,U).astype(F32)        # [36, tiles, Cout]
        M=M.reshape(6,6,th,tw,Cout)
        Y=np.einsum('ia,abtuc,jb->tiujc',AT,M,AT,optimize=True).astype(F32)   # [th,4,tw,4,Cout]
        out[bi]=Y.reshape(th*4,tw*4,Cout)
    out=out[:,:H,:W]
    if b is not None: out=out+b.astype(F32)
    return out.astype(F32)
O.conv3x3=conv_f43
def pre_ratio(got,ref):
    err=np.abs(got-ref); return float((err/(1e-4+1e-3*np.abs(ref))).max()), float(err.max())
w=pkg.synthetic_weights(0)
for case in sys.argv[1:]:
    if case=="global_a":
        g=C.load_golden("global_a")
        style,frames,ids,tid=C.golden_inputs(pkg,g)
        o=O.Stylization(w); o.set_state(g["state"])
        padded=O.reflect_pad(frames[tid],192,192)
        for mode in ("none","all"):
            MODE["layers"]=mode
            pre=o.transfer(padded,return_preclamp=True)[0]
            print(case, mode, "pre worst ratio %.3f max|d| %.2e"%pre_ratio(pre,g["pre"]))
    if case=="real_default":
        g=C.load_golden("real_default")
        o=O.Stylization(w); o.set_state(g["state"])
        padded=O.reflect_pad(C.decode_png(g["frame12_png"]),576,1152)
        for mode in ("none","all"):
            MODE["layers"]=mode
            pre=o.transfer(padded,return_preclamp=True)[0][64:500,64:1088]
            print(case, mode, "pre worst ratio %.3f max|d| %.2e (grid)"%pre_ratio(pre[::4,::4],g["pre_grid"]))
