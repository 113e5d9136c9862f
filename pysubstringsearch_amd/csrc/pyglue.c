/* pyglue.c -- CPython helper for the ctypes shim: turns a packed result
 * (entry bytes + offsets, as returned by pss_result_bytes / pss_result_offsets,
 * include/pss.h) into a Python list in one C loop.  The reference does the same
 * step natively (pyo3 Vec<&str> -> list[str], src/lib.rs:284-286); doing it with
 * a Python-level slice loop costs ~3x more per entry and dominates hit-heavy
 * batches.  Host-side marshalling only: no search logic lives here. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

/* Entries are ASCII far more often than not; for those, a 1-byte-per-character str is the bytes
 * themselves: PyUnicode_New + memcpy skips the UTF-8 decoder's state machine (~20 % of the time per
 * entry, and the list is what bounds hit-heavy batches). */
static inline int all_ascii(const unsigned char *p, size_t len)
{
    uint64_t acc = 0;
    size_t i = 0;
    for (; i + 8 <= len; i += 8) {
        uint64_t w;
        memcpy(&w, p + i, 8);
        acc |= w;
    }
    for (; i < len; ++i) acc |= p[i];
    return (acc & 0x8080808080808080ull) == 0;
}

/* entries_to_list(bytes_addr: int, offsets_addr: int, n: int, as_str: bool) -> list */
static PyObject *entries_to_list(PyObject *self, PyObject *args)
{
    unsigned long long bytes_addr, off_addr;
    Py_ssize_t n;
    int as_str;
    (void)self;
    if (!PyArg_ParseTuple(args, "KKnp", &bytes_addr, &off_addr, &n, &as_str)) return NULL;
    const char *base = (const char *)(uintptr_t)bytes_addr;
    const uint64_t *off = (const uint64_t *)(uintptr_t)off_addr;
    PyObject *list = PyList_New(n);
    if (!list) return NULL;
    for (Py_ssize_t i = 0; i < n; ++i) {
        const char *p = base + off[i];
        const Py_ssize_t len = (Py_ssize_t)(off[i + 1] - off[i]);
        PyObject *o;
        if (!as_str) {
            o = PyBytes_FromStringAndSize(p, len);
        } else if (all_ascii((const unsigned char *)p, (size_t)len)) {
            o = PyUnicode_New(len, 127);
            if (o && len) memcpy(PyUnicode_1BYTE_DATA(o), p, (size_t)len);
        } else {
            o = PyUnicode_DecodeUTF8(p, len, "strict");
        }
        if (!o) {
            Py_DECREF(list);
            return NULL;
        }
        PyList_SET_ITEM(list, i, o);
    }
    return list;
}

/* pack_queries(seq_of_bytes) -> (blob: bytes, offsets: bytes holding (n+1) little-endian u64)
 * One C loop instead of b''.join + a numpy cumsum; raises TypeError on a non-bytes item. */
static PyObject *pack_queries(PyObject *self, PyObject *arg)
{
    (void)self;
    PyObject *seq = PySequence_Fast(arg, "expected a sequence of bytes");
    if (!seq) return NULL;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    PyObject **items = PySequence_Fast_ITEMS(seq);
    Py_ssize_t total = 0;
    for (Py_ssize_t i = 0; i < n; ++i) {
        if (!PyBytes_Check(items[i])) {
            Py_DECREF(seq);
            PyErr_SetString(PyExc_TypeError, "queries must be bytes");
            return NULL;
        }
        total += PyBytes_GET_SIZE(items[i]);
    }
    PyObject *blob = PyBytes_FromStringAndSize(NULL, total);
    PyObject *offs = PyBytes_FromStringAndSize(NULL, (n + 1) * (Py_ssize_t)sizeof(uint64_t));
    if (!blob || !offs) {
        Py_XDECREF(blob);
        Py_XDECREF(offs);
        Py_DECREF(seq);
        return NULL;
    }
    char *bp = PyBytes_AS_STRING(blob);
    uint64_t *op = (uint64_t *)PyBytes_AS_STRING(offs);
    uint64_t pos = 0;
    for (Py_ssize_t i = 0; i < n; ++i) {
        const Py_ssize_t l = PyBytes_GET_SIZE(items[i]);
        op[i] = pos;
        memcpy(bp + pos, PyBytes_AS_STRING(items[i]), (size_t)l);
        pos += (uint64_t)l;
    }
    op[n] = pos;
    Py_DECREF(seq);
    return Py_BuildValue("(NN)", blob, offs);
}

/* u64_list(addr: int, n: int) -> list[int] */
static PyObject *u64_list(PyObject *self, PyObject *args)
{
    unsigned long long addr;
    Py_ssize_t n;
    (void)self;
    if (!PyArg_ParseTuple(args, "Kn", &addr, &n)) return NULL;
    const uint64_t *p = (const uint64_t *)(uintptr_t)addr;
    PyObject *list = PyList_New(n);
    if (!list) return NULL;
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject *o = PyLong_FromUnsignedLongLong(p[i]);
        if (!o) {
            Py_DECREF(list);
            return NULL;
        }
        PyList_SET_ITEM(list, i, o);
    }
    return list;
}

static PyMethodDef methods[] = {
    {"pack_queries", pack_queries, METH_O, "sequence of bytes -> (blob, u64 offsets as bytes)"},
    {"u64_list", u64_list, METH_VARARGS, "(address, n) -> list of ints"},
    {"entries_to_list", entries_to_list, METH_VARARGS, "packed (bytes, offsets) -> list of str / bytes"},
    {NULL, NULL, 0, NULL},
};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_pssglue", "result marshalling for pysubstringsearch_amd", -1,
                                    methods, NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__pssglue(void) { return PyModule_Create(&moddef); }
