import numpy as np
from scipy.special import erf
from scipy.optimize import least_squares
f32=np.float32
def model(params, z, n):
    p = params[0]; b = params[1:]
    s = p*z/(1.0+p*z)
    Q = np.zeros_like(z)
    for k in range(n,0,-1):
        Q = (Q + b[k-1]) * s
    return 1.0 - (1.0+Q)*np.exp(-z*z)
def fit(n, p0=0.33):
    z = np.linspace(0,6,20001)
    target = erf(z)
    x0 = np.zeros(n+1); x0[0]=p0
    s = p0*z/(1+p0*z); A = np.stack([s**k*np.exp(-z*z) for k in range(1,n+1)],1)
    b,_r,_,_ = np.linalg.lstsq(A, 1-target-np.exp(-z*z), rcond=None); x0[1:]=b
    w = np.ones_like(z); best=None
    for it in range(80):
        r = least_squares(lambda q: w*(model(q,z,n)-target), x0, xtol=1e-15, ftol=1e-15, gtol=1e-15)
        x0 = r.x
        err = np.abs(model(x0,z,n)-target); m = err.max()
        if best is None or m<best[0]: best=(m,x0.copy())
        w = w*(1+2*err/m); w/=w.mean()
    return best
def fma(a,b,c): return (a.astype(np.float64)*b.astype(np.float64)+c.astype(np.float64)).astype(f32)
def sim(n,x):
    p=f32(x[0]); b=[f32(v) for v in x[1:]]
    rng=np.random.default_rng(0)
    u=np.concatenate([rng.uniform(-8,8,2000000), rng.normal(0,1,2000000), np.linspace(-1e-3,1e-3,100001)]).astype(f32)
    pz=(np.abs(u)*f32(np.float64(x[0])*0.70710678118654752)).astype(f32)
    den=(pz+f32(1)).astype(f32)
    t=(f32(1)/den).astype(f32)
    s=(pz*t).astype(f32)
    arg=((u*u).astype(f32)*f32(-0.72134752044448170)).astype(f32)
    e=np.exp2(arg.astype(np.float64)).astype(f32)
    Q=np.full_like(s,b[-1])
    for k in range(n-2,-1,-1): Q=fma(Q,s,np.full_like(s,b[k]))
    Q=fma(Q,s,np.ones_like(s))
    erfabs=fma(-Q,e,np.ones_like(s))
    h=fma((f32(0.5)*np.abs(u)).astype(f32),erfabs,(f32(0.5)*u).astype(f32))
    ud=u.astype(np.float64)
    h_ref=0.5*ud*(1+erf(ud/np.sqrt(2)))
    er32=erf((u*f32(0.70710678118654752)).astype(f32).astype(np.float64)).astype(f32)
    h32=((f32(0.5)*u).astype(f32)*(f32(1)+er32).astype(f32)).astype(f32)
    cdf=(f32(0.5)+np.copysign((f32(0.5)*erfabs).astype(f32),u)).astype(f32)
    pdf=(e*f32(0.39894228040143268)).astype(f32)
    g=fma(u,pdf,cdf)
    g_ref=0.5*(1+erf(ud/np.sqrt(2)))+ud*np.exp(-ud*ud/2)/np.sqrt(2*np.pi)
    g32=(f32(0.5)*(f32(1)+er32)+u*(np.exp((f32(-0.5)*u*u).astype(np.float64)).astype(f32)*f32(0.39894228040143268))).astype(f32)
    print(n,"erf abs",np.abs(erfabs-erf(np.abs(ud)/np.sqrt(2))).max(),"gelu abs",np.abs(h-h_ref).max(),"(ref32",np.abs(h32-h_ref).max(),
      ") rel|u|",(np.abs(h-h_ref)/np.maximum(np.abs(ud),1e-30)).max(),"(ref32",(np.abs(h32-h_ref)/np.maximum(np.abs(ud),1e-30)).max(),
      ") gelu' abs",np.abs(g-g_ref).max(),"(ref32",np.abs(g32-g_ref).max(),")")
for n in (5,6,7):
    m,x=fit(n)
    print(n,"exact maxerr",m,"p",repr(float(x[0])),"b",[float(v) for v in x[1:]])
    sim(n,x)
