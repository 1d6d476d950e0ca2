import ctypes, numpy as np
libm = ctypes.CDLL("libm.so.6")
libm.sinf.argtypes=[ctypes.c_float]; libm.sinf.restype=ctypes.c_float
libm.cosf.argtypes=[ctypes.c_float]; libm.cosf.restype=ctypes.c_float
H=float.fromhex
T=[dict(sign=[1.0,-1.0,-1.0,1.0], hpi_inv=H('0x1.45f306dc9c883p+23'), hpi=H('0x1.921fb54442d18p+0'),
        c0=1.0, c1=H('-0x1.ffffffd0c621cp-2'), c2=H('0x1.55553e1068f19p-5'), c3=H('-0x1.6c087e89a359dp-10'), c4=H('0x1.99343027bf8c3p-16'),
        s1=H('-0x1.555545995a603p-3'), s2=H('0x1.1107605230bc4p-7'), s3=H('-0x1.994eb3774cf24p-13'))]
T.append(dict(T[0])); 
for k in ('c0','c1','c2','c3','c4'): T[1][k] = -T[0][k]
def poly(x, x2, p, n):
    if (n & 1) == 0:
        x3 = x*x2; s1 = p['s2'] + x2*p['s3']; x5 = x3*x2; s = x + x3*p['s1']; return np.float32(s + x5*s1)
    else:
        x4 = x2*x2; c2 = p['c3'] + x2*p['c4']; c1 = p['c0'] + x2*p['c1']; x6 = x4*x2; c = c1 + x4*p['c2']; return np.float32(c + x6*c2)
def abstop12(f): return (np.float32(f).view(np.uint32) >> 20) & 0x7ff
PIO4 = abstop12(np.float32(0.7853981633974483)); T12=abstop12(np.float32(2.0**-12)); T120=abstop12(np.float32(120.0))
def reduce_fast(x, p):
    r = x*p['hpi_inv']
    n = (int(np.int32(r)) + 0x800000) >> 24
    return x - n*p['hpi'], n
def sinf(y):
    y=np.float32(y); x=float(y); p=T[0]
    if abstop12(y) < PIO4:
        s=x*x
        if abstop12(y) < T12: return y
        return poly(x,s,p,0)
    elif abstop12(y) < T120:
        x,n = reduce_fast(x,p); s=p['sign'][n&3]
        if n&2: p=T[1]
        return poly(x*s, x*x, p, n)
    return None
def cosf(y):
    y=np.float32(y); x=float(y); p=T[0]
    if abstop12(y) < PIO4:
        x2=x*x
        if abstop12(y) < T12: return np.float32(1.0)
        return poly(x,x2,p,1)
    elif abstop12(y) < T120:
        x,n = reduce_fast(x,p); s=p['sign'][n&3]
        if n&2: p=T[1]
        return poly(x*s, x*x, p, n^1)
    return None
if __name__ == "__main__":
    rng=np.random.default_rng(0)
    bad=0; tot=0
    xs = np.concatenate([rng.uniform(-0.79,2.36,300000), rng.uniform(-119,119,200000), rng.uniform(-1e-3,1e-3,20000)]).astype(np.float32)
    worst=[]
    for v in xs:
        a=sinf(v); b=cosf(v)
        la=np.float32(libm.sinf(float(v))); lb=np.float32(libm.cosf(float(v)))
        tot+=2
        if a.view(np.uint32)!=la.view(np.uint32): bad+=1; worst.append(('sin',v,a,la))
        if b.view(np.uint32)!=lb.view(np.uint32): bad+=1; worst.append(('cos',v,b,lb))
    print("port vs libm mismatches:", bad, "of", tot)
    print(worst[:5])
