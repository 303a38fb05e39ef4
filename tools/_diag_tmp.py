import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, scipy.stats, warnings
import polar_amd
T = np.load('tests/golden/construction_tables.npz')
for key in [str(k) for k in T['keys']]:
    ref = T[key + '/counts'].astype(np.float64); snr, runs = float(T[key+'/meta'][0]), int(T[key+'/meta'][1])
    const = key.split('_')[0]
    got = polar_amd.mc_construction(10, snr, runs, const, seed=77).astype(np.float64)
    tol = 6.0 * np.sqrt(ref + got + 1.0) + 3.0
    bad = np.nonzero(np.abs(got - ref) > tol)[0]
    big = (ref + got) >= 20
    with np.errstate(all='ignore'):
        chi2 = float((((got - ref) ** 2) / (got + ref))[big].sum())
    rho = scipy.stats.spearmanr(got, ref).correlation
    z = (got-ref)/np.sqrt(got+ref+1)
    print(key, 'bad', bad.size, 'chi2/dof %.3f (%d)' % (chi2/big.sum(), big.sum()), 'rho %.5f' % rho, 'sum ratio %.5f' % (got.sum()/ref.sum()), 'max|z| %.2f' % np.abs(z).max(), 'mean z %.3f' % z[big].mean())
