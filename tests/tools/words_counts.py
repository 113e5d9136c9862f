import sys, numpy as np
sys.path.insert(0, '.')
import tests.test_fullsize_gpu as T
reader, texts, sas = T._corpus_reader(29, 15, keep_sa=False, kind='words')
for qmin in (10, 12, 16):
    q = T._batch(texts, 100000, qmin=qmin)
    c = np.array(reader.count_multiple_bytes(q), dtype=np.int64)
    lens = np.array([len(x) for x in q])
    print('qmin', qmin, 'total entries', int(c.sum()), 'max', int(c.max()), 'mean', float(c.mean()), 'p50/p90/p99/p999', np.percentile(c, [50, 90, 99, 99.9]).tolist(),
          'queries with > 1e5 hits', int((c > 100000).sum()), 'with > 1e4', int((c > 10000).sum()))
    big = np.argsort(-c)[:5]
    print('   largest:', [(q[i], int(c[i])) for i in big])
