#!/usr/bin/env python3
"""CPU only, build container only (imports the reference from /root/reference through tests/golden/refshim.py): the PRODUCT'S
host logic (cna_amd.tl.association on the test double tests/fake_engine.py: numpy in place of the kernels) against the
reference itself on random small problems with messy sample-level inputs -- shuffled and
partial indices, NaNs, unused categories, batches, donor groups, custom ks / ridges / max_frac_pcs, few samples.  Same
exception type AND message, or the same numbers at 1e-7 (the double computes in float64 throughout).
    python tools/fuzz_hostlogic_vs_reference.py [seconds=300] [seed=0]"""
import os, sys, time, warnings, io, contextlib, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
warnings.simplefilter('ignore')
import numpy as np, pandas as pd
from cna_amd import synth
from oracle import cna_oracle as orc
import cna_amd as cna_new
from fake_engine import FakeEngine
import refshim
ref = refshim.load_reference()

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def relerr(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    if a.shape != b.shape:
        return np.inf
    d = np.abs(a - b)
    return float(np.nanmax(d) / max(np.nanmax(np.abs(b)), 1e-300)) if d.size else 0.0


t_end = time.time() + budget
done = both = 0
fails = {}
while time.time() < t_end:
    n = int(rs.choice([600, 900, 1500]))
    N = int(rs.choice([8, 11, 14, 20, 26, 40, 64]))
    opts = dict(k=int(rs.choice([8, 15])), seed=int(rs.randint(1 << 30)), graph_dtype=rs.choice([np.float32, np.float64]),
                cluster_sorted=bool(rs.rand() < 0.5), sid_kind=str(rs.choice(['int', 'str', 'cat'])),
                n_covs=int(rs.choice([0, 0, 1, 2])), n_batches=int(rs.choice([0, 0, 3, 5, 9])), builder='cpu')
    data, meta = synth.make_dataset(n, N, **opts)
    y, covs, batches, donor = meta['y'].copy(), meta['covs'], meta['batches'], None
    kw = dict(nsteps=rs.choice([None, 2, 3]), Nnull=int(rs.choice([20, 50, 101])), seed=int(rs.randint(1000)))
    kw['nsteps'] = None if kw['nsteps'] is None else int(kw['nsteps'])
    tag = ['N%d' % N, opts['sid_kind']]
    if rs.rand() < 0.25:
        y.iloc[int(rs.randint(N))] = np.nan; tag.append('ynan')
    if covs is not None and rs.rand() < 0.25:
        covs = covs.copy(); covs.iloc[int(rs.randint(N)), 0] = np.nan; tag.append('covnan')
    if batches is not None and rs.rand() < 0.15:
        batches = batches.astype(float).copy(); batches.iloc[int(rs.randint(N))] = np.nan; tag.append('batchnan')
    if opts['sid_kind'] == 'cat' and rs.rand() < 0.4:
        col = data.obs['id']; codes = np.asarray(col.cat.codes).copy(); codes[codes == 2] = 3
        data.obs['id'] = pd.Categorical.from_codes(codes, categories=col.cat.categories); tag.append('unused')
    if rs.rand() < 0.3:                                    # every sample-level input in its own order
        perm = rs.permutation(len(y)); y = y.iloc[perm]; tag.append('yperm')
        if covs is not None and rs.rand() < 0.5:
            covs = covs.iloc[rs.permutation(len(covs))]; tag.append('covperm')
        if batches is not None and rs.rand() < 0.5:
            batches = batches.iloc[rs.permutation(len(batches))]; tag.append('bperm')
    if rs.rand() < 0.15 and opts['sid_kind'] == 'int':
        y = pd.concat([y, pd.Series([0.3, -1.2], index=pd.Index([5000, 5001]))]); tag.append('yextra')
        if covs is not None:
            covs = pd.concat([covs, pd.DataFrame(np.zeros((2, covs.shape[1])), index=[5000, 5001], columns=covs.columns)])
        if batches is not None:
            batches = pd.concat([batches, pd.Series([0, 1], index=[5000, 5001])])
    if batches is None and rs.rand() < 0.15 and N >= 14 and 'yperm' not in tag and 'yextra' not in tag:
        donor = pd.Series(np.arange(len(y)) // 2, index=y.index); y[:] = np.repeat(rs.randn((len(y) + 1) // 2), 2)[:len(y)]; tag.append('donor')
    if rs.rand() < 0.15:
        kw['force_permute_all'] = True; tag.append('fpa')
    if rs.rand() < 0.2:
        kw['ks'] = [int(v) for v in rs.choice([1, 2, 3, 4, 6], size=2, replace=False)]; tag.append('ks')
    if rs.rand() < 0.15:
        kw['max_frac_pcs'] = float(rs.choice([0.05, 0.3, 0.5])); tag.append('maxfrac')
    if batches is not None and rs.rand() < 0.2:
        kw['ridges'] = [float(v) for v in rs.choice([1e3, 10.0, 1.0, 0.0], size=2, replace=False)]; tag.append('ridges')
    if N < 10 and rs.rand() < 0.6:
        kw['allow_low_sample_size'] = True; tag.append('lowN')
    if rs.rand() < 0.1:
        kw['local_test'] = False; tag.append('nolocal')
    if rs.rand() < 0.1:
        kw['Nnull'] = int(rs.choice([1001, 1200])); tag.append('bigNnull')
    if covs is not None and rs.rand() < 0.1:
        covs = covs.copy(); covs.iloc[:, 0] = 1.0; tag.append('constcov')
    if batches is not None and rs.rand() < 0.15:
        batches = batches.map(lambda v: 'b%s' % v if v == v else v); tag.append('strbatch')
    if rs.rand() < 0.1:
        y = y.round().astype(int) if not y.isna().any() else y; tag.append('inty')
    if donor is not None and rs.rand() < 0.3:
        y = y.copy(); y.iloc[0] += 1.0; tag.append('donor_inconsistent')
    progress = rs.rand() < 0.3
    if progress:
        tag.append('progress')
    d2 = type('D', (), {})(); d2.obs = data.obs.copy(); d2.obsp = data.obsp; d2.uns = {}
    a = b = ea = eb = None
    try:
        buf_a = io.StringIO()
        with contextlib.redirect_stdout(buf_a):
            a = ref.tl.association(d2, y, 'id', covs=covs, batches=batches, donorids=donor, return_full=True, show_progress=progress, **kw)
    except Exception as e:                       # noqa: BLE001
        ea = e
    d3 = type('D', (), {})(); d3.obs = data.obs.copy(); d3.obsp = data.obsp; d3.uns = {}
    try:
        buf_b = io.StringIO()
        with contextlib.redirect_stdout(buf_b):
            b = cna_new.tl.association(d3, y, 'id', covs=covs, batches=batches, donorids=donor, return_full=True,
                                       engine=FakeEngine(), show_progress=progress, **kw)
    except Exception as e:                       # noqa: BLE001
        eb = e
    # cna.tl.nam on the same inputs (the public function: frame layout, QC mask, self weight)
    if rs.rand() < 0.3:
        sw = float(rs.choice([1, 1, 0.5, 2]))
        na = nb_ = en = enb = None
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                na = ref.tl.nam(d2, 'id', batches=batches, nsteps=kw['nsteps'], self_weight=sw)
        except Exception as e:                   # noqa: BLE001
            en = e
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                nb_ = cna_new.tl.nam(d3, 'id', batches=batches, nsteps=kw['nsteps'], self_weight=sw, engine=FakeEngine())
        except Exception as e:                   # noqa: BLE001
            enb = e
        try:
            if en is not None or enb is not None:
                assert en is not None and enb is not None and type(en) is type(enb) and str(en) == str(enb), ('tl.nam exceptions', repr(en)[:100], repr(enb)[:100])
            else:
                assert np.array_equal(np.asarray(na[1]), np.asarray(nb_[1])), 'tl.nam keep'
                assert list(na[0].index) == list(nb_[0].index) and list(na[0].columns) == list(nb_[0].columns), 'tl.nam labels'
                assert relerr(na[0].values, nb_[0].values) < 1e-6, ('tl.nam values', relerr(na[0].values, nb_[0].values))
        except Exception as exc:                 # noqa: BLE001
            what = str(exc.args[0] if exc.args else exc)[:200]
            fails.setdefault(what.split(',')[0][:50], []).append((done, n, tag + ['sw%s' % sw], kw, opts['seed'], what))
    done += 1
    if os.environ.get('FUZZ_DEBUG_CASE') and done == int(os.environ['FUZZ_DEBUG_CASE']):
        print('case', done, n, N, opts, kw, tag)
        print('y index', list(y.index)[:12], 'nan at', list(y.index[y.isna()]))
        print('covs', None if covs is None else (covs.shape, list(covs.index)[:8]), 'batches', None if batches is None else dict(batches.value_counts()))
        if ea or eb:
            print('raised', repr(ea), repr(eb))
            import traceback as _tb
            if eb is not None:
                _tb.print_exception(type(eb), eb, eb.__traceback__, limit=-4)
        else:
            print('p', a.p, b.p, 'k', a.k, b.k, 'r', a.r, b.r, 'ks', a.ks, b.ks)
            d = np.abs(np.asarray(a.nullminps) - np.asarray(b.nullminps))
            print('nullminps max diff', d.max(), 'at', int(d.argmax()), a.nullminps[d.argmax()], b.nullminps[d.argmax()], 'count > 1e-9:', int((d > 1e-9).sum()))
            print('yresid diff', np.abs(a.yresid - b.yresid.values if hasattr(b.yresid, 'values') else a.yresid - b.yresid).max())
            print('svs', a.namresid_svs.values[:5], b.namresid_svs.values[:5])
            T_ = min(len(a.fdrs), len(b.fdrs)); dd = a.fdrs.num_detected.values[:T_] - b.fdrs.num_detected.values[:T_]
            print('len fdrs', len(a.fdrs), len(b.fdrs), 'num_detected diffs at', np.flatnonzero(dd)[:10], dd[np.flatnonzero(dd)[:10]], 'thr', a.fdrs.threshold.values[np.flatnonzero(dd)[:3]], b.fdrs.threshold.values[np.flatnonzero(dd)[:3]])
            nc_a = np.sort(np.abs(a.ncorrs.values)); nc_b = np.sort(np.abs(b.ncorrs.values)); print('max |ncorrs| rel diff', np.abs(nc_a - nc_b).max() / nc_a.max())
        break
    try:
        if ea is not None or eb is not None:
            assert ea is not None and eb is not None, ('one side raised', repr(ea)[:120], repr(eb)[:120])
            assert type(ea) is type(eb) and str(ea) == str(eb), ('exceptions differ', repr(ea)[:120], repr(eb)[:120])
            both += 1
            continue
        import re as _re
        mask = lambda t: _re.sub(r'[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?|nan|inf', '#', t)
        assert mask(buf_a.getvalue()) == mask(buf_b.getvalue()), ('stdout', buf_a.getvalue()[-300:], buf_b.getvalue()[-300:])
        assert int(a.k) == int(b.k) and abs(a.p - b.p) < 1e-12, ('k / p', a.k, b.k, a.p, b.p)
        assert np.array_equal(a.kept, b.kept), ('kept', int(a.kept.sum()), int(b.kept.sum()))
        assert list(a.nam.index) == list(b.nam.index), 'sample order of res.nam'
        assert relerr(a.nam.values, b.nam.values) < 1e-6, ('nam', relerr(a.nam.values, b.nam.values))
        assert relerr(a.ncorrs.values, b.ncorrs.values) < 1e-6, 'ncorrs'
        assert relerr(a.nullminps, b.nullminps) < 1e-5, 'nullminps'
        T = min(len(a.fdrs), len(b.fdrs))
        assert np.array_equal(a.fdrs.num_detected.values[:T], b.fdrs.num_detected.values[:T]), 'num_detected'
        assert relerr(d2.obs['coef'].values, d3.obs['coef'].values) < 1e-6, 'obs coef'
        for field in ('yresid', 'M', 'namresid'):
            av, bv = getattr(a, field), getattr(b, field)
            assert list(getattr(av, 'index', [])) == list(getattr(bv, 'index', [])), field + ' index'
    except Exception as exc:                     # noqa: BLE001
        what = str(exc.args[0] if exc.args else exc)[:200]
        fails.setdefault(what.split(',')[0][:50], []).append((done, n, tag, kw, opts['seed'], what))
for key, items in fails.items():
    print('== %d x %s' % (len(items), key))
    for it in items[:5]:
        print('   case %d n=%d %s %r seed=%d\n      %s' % it)
print('%d cases in %.0f s: %d disagreements, %d raised the same exception type on both sides' % (done, budget, sum(len(v) for v in fails.values()), both))
