/* aix_pyfast.c — CPython helper for the list[str] surface of the reference's API (get_tf_values(list[str]) -> list[int],
 * python_wrapper.cpp:653-664 takes std::vector<std::string> and returns std::vector<uint32_t>; pybind11 converts both ways, GIL held).
 * At 10^8 queries the two conversions ARE the call: the GPU answers the batch in 50 ms, a Python loop over the items takes seconds.
 *
 *   join_fixed(seq, k[, threads])   -> bytes of len(seq) * k, or None when an item is not a k-character ASCII str / k-byte bytes.
 *                                      The items' character buffers are read by `threads` worker threads (the calling thread keeps the GIL
 *                                      and waits: nothing can mutate or free the items meanwhile, and the workers touch no Python API —
 *                                      they only read the immutable object headers and buffers).
 *   join_fixed_into(seq, k, buf[, threads]) -> the same into a caller's writable buffer (the wrapper's pinned staging, kept between calls).
 *   ascii_view(str)                 -> the characters of an ASCII str as a read-only memoryview (a batch handed over as ONE joined str is
 *                                      looked up straight from the str's own buffer).
 *   u32_list(buffer, n[, threads])  -> list[int] of the first n uint32 of a buffer. Values <= 256 are CPython's cached small ints: the
 *                                      workers store their addresses straight into the list and tally how often each one was used, the
 *                                      calling thread then adds the tallies to the reference counts; larger values (rare for term
 *                                      frequencies) are boxed afterwards by the calling thread.
 * Host glue only; when the module is missing the pure-Python path is used (same answers). */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAX_THREADS 64

static int pick_threads(Py_ssize_t n, long asked) {
    long t = asked > 0 ? asked : 16;
    if (t > MAX_THREADS) t = MAX_THREADS;
    if (n < 200000) t = 1;                              /* a small batch is not worth a thread start */
    return (int)t;
}

/* ---- join_fixed ------------------------------------------------------------------------------------------------------------ */
typedef struct {
    PyObject** items;
    Py_ssize_t lo, hi, k;
    char* dst;
    int bad;
} join_job;

static void* join_worker(void* p) {
    join_job* j = (join_job*)p;
    const Py_ssize_t k = j->k;
    for (Py_ssize_t i = j->lo; i < j->hi; ++i) {
        PyObject* it = j->items[i];
        const char* src = NULL;
        if (i + 16 < j->hi) {                           /* the items lie anywhere on the heap: ask for the one 16 ahead now (header and characters: two lines) */
            const char* nx = (const char*)j->items[i + 16];
            __builtin_prefetch(nx, 0, 0);
            __builtin_prefetch(nx + 64, 0, 0);
        }
        if (PyUnicode_Check(it)) {                      /* macros only: type flags and the compact-ASCII header are read, nothing is called */
            if (PyUnicode_IS_READY(it) && PyUnicode_IS_COMPACT_ASCII(it) && PyUnicode_GET_LENGTH(it) == k) src = (const char*)(((PyASCIIObject*)it) + 1);
        } else if (PyBytes_Check(it)) {
            if (PyBytes_GET_SIZE(it) == k) src = PyBytes_AS_STRING(it);
        }
        if (!src) { j->bad = 1; return NULL; }
        memcpy(j->dst + i * k, src, (size_t)k);
    }
    return NULL;
}

static PyObject* join_fixed(PyObject* self, PyObject* args) {
    PyObject* seq;
    Py_ssize_t k;
    long threads = 0;
    (void)self;
    if (!PyArg_ParseTuple(args, "On|l", &seq, &k, &threads)) return NULL;
    if (k <= 0) Py_RETURN_NONE;
    PyObject* fast = PySequence_Fast(seq, "join_fixed: expected a sequence");
    if (!fast) return NULL;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
    if (n > PY_SSIZE_T_MAX / k) { Py_DECREF(fast); Py_RETURN_NONE; }
    PyObject* out = PyBytes_FromStringAndSize(NULL, n * k);
    if (!out) { Py_DECREF(fast); return NULL; }
    const int nt = pick_threads(n, threads);
    join_job jobs[MAX_THREADS];
    pthread_t th[MAX_THREADS];
    int started[MAX_THREADS];
    PyObject** items = PySequence_Fast_ITEMS(fast);
    for (int t = 0; t < nt; ++t) {
        jobs[t].items = items; jobs[t].k = k; jobs[t].dst = PyBytes_AS_STRING(out); jobs[t].bad = 0;
        jobs[t].lo = n / nt * t + (t < n % nt ? t : n % nt);
        jobs[t].hi = jobs[t].lo + n / nt + (t < n % nt ? 1 : 0);
        started[t] = 0;
    }
    /* the GIL stays with this thread while the workers run: the list and its items cannot change under them */
    for (int t = 1; t < nt; ++t) started[t] = pthread_create(&th[t], NULL, join_worker, &jobs[t]) == 0;
    join_worker(&jobs[0]);
    int bad = jobs[0].bad;
    for (int t = 1; t < nt; ++t) {
        if (started[t]) pthread_join(th[t], NULL); else join_worker(&jobs[t]);
        bad |= jobs[t].bad;
    }
    Py_DECREF(fast);
    if (bad) { Py_DECREF(out); Py_RETURN_NONE; }
    return out;
}

/* join_fixed_into(seq, k, buffer[, threads]) -> True, or None when an item is not a k-character ASCII str / k-byte bytes (buffer contents
 * then undefined). `buffer` is a writable buffer of at least len(seq) * k bytes — the wrapper's pinned staging, written in place. */
static PyObject* join_fixed_into(PyObject* self, PyObject* args) {
    PyObject* seq;
    Py_ssize_t k;
    Py_buffer view;
    long threads = 0;
    (void)self;
    if (!PyArg_ParseTuple(args, "Onw*|l", &seq, &k, &view, &threads)) return NULL;
    if (k <= 0) { PyBuffer_Release(&view); Py_RETURN_NONE; }
    PyObject* fast = PySequence_Fast(seq, "join_fixed_into: expected a sequence");
    if (!fast) { PyBuffer_Release(&view); return NULL; }
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
    if (n > PY_SSIZE_T_MAX / k || view.len < n * k) {
        Py_DECREF(fast); PyBuffer_Release(&view);
        PyErr_SetString(PyExc_ValueError, "join_fixed_into: buffer shorter than len(seq) * k bytes");
        return NULL;
    }
    const int nt = pick_threads(n, threads);
    join_job jobs[MAX_THREADS];
    pthread_t th[MAX_THREADS];
    int started[MAX_THREADS];
    PyObject** items = PySequence_Fast_ITEMS(fast);
    for (int t = 0; t < nt; ++t) {
        jobs[t].items = items; jobs[t].k = k; jobs[t].dst = (char*)view.buf; jobs[t].bad = 0;
        jobs[t].lo = n / nt * t + (t < n % nt ? t : n % nt);
        jobs[t].hi = jobs[t].lo + n / nt + (t < n % nt ? 1 : 0);
        started[t] = 0;
    }
    for (int t = 1; t < nt; ++t) started[t] = pthread_create(&th[t], NULL, join_worker, &jobs[t]) == 0;      /* GIL held by this thread throughout, as in join_fixed */
    join_worker(&jobs[0]);
    int bad = jobs[0].bad;
    for (int t = 1; t < nt; ++t) {
        if (started[t]) pthread_join(th[t], NULL); else join_worker(&jobs[t]);
        bad |= jobs[t].bad;
    }
    Py_DECREF(fast);
    PyBuffer_Release(&view);
    if (bad) Py_RETURN_NONE;
    Py_RETURN_TRUE;
}

/* ---- ascii_view ------------------------------------------------------------------------------------------------------------ */
/* ascii_view(s) -> read-only memoryview over the characters of an ASCII str (no copy, no encode pass); None for any other str. The view
 * borrows the str's own buffer: the caller keeps the str alive while it uses the view. */
static PyObject* ascii_view(PyObject* self, PyObject* arg) {
    (void)self;
    if (!PyUnicode_Check(arg) || PyUnicode_READY(arg) != 0 || !PyUnicode_IS_COMPACT_ASCII(arg)) Py_RETURN_NONE;
    return PyMemoryView_FromMemory((char*)(((PyASCIIObject*)arg) + 1), PyUnicode_GET_LENGTH(arg), PyBUF_READ);
}

/* ---- u32_list -------------------------------------------------------------------------------------------------------------- */
#define NSMALL 257
static PyObject* g_small[NSMALL];                       /* 0 .. 256: CPython's cached small ints (one strong reference each, held by the module) */

typedef struct {
    const uint32_t* src;
    PyObject** dst;
    Py_ssize_t lo, hi;
    Py_ssize_t tally[NSMALL];
    Py_ssize_t big;
} list_job;

static void* list_worker(void* p) {
    list_job* j = (list_job*)p;
    memset(j->tally, 0, sizeof(j->tally));
    j->big = 0;
    for (Py_ssize_t i = j->lo; i < j->hi; ++i) {
        const uint32_t v = j->src[i];
        if (v < NSMALL) { j->dst[i] = g_small[v]; ++j->tally[v]; }
        else { j->dst[i] = NULL; ++j->big; }
    }
    return NULL;
}

static PyObject* u32_list(PyObject* self, PyObject* args) {
    Py_buffer view;
    Py_ssize_t n;
    long threads = 0;
    (void)self;
    if (!PyArg_ParseTuple(args, "y*n|l", &view, &n, &threads)) return NULL;
    if (n < 0 || (Py_ssize_t)(view.len / 4) < n) { PyBuffer_Release(&view); PyErr_SetString(PyExc_ValueError, "u32_list: buffer shorter than n uint32"); return NULL; }
    PyObject* list = PyList_New(n);
    if (!list) { PyBuffer_Release(&view); return NULL; }
    if (n == 0) { PyBuffer_Release(&view); return list; }
    const int nt = pick_threads(n, threads);
    list_job* jobs = (list_job*)malloc(sizeof(list_job) * (size_t)nt);
    if (!jobs) { PyBuffer_Release(&view); Py_DECREF(list); return PyErr_NoMemory(); }
    pthread_t th[MAX_THREADS];
    int started[MAX_THREADS];
    PyObject** dst = ((PyListObject*)list)->ob_item;
    for (int t = 0; t < nt; ++t) {
        jobs[t].src = (const uint32_t*)view.buf; jobs[t].dst = dst;
        jobs[t].lo = n / nt * t + (t < n % nt ? t : n % nt);
        jobs[t].hi = jobs[t].lo + n / nt + (t < n % nt ? 1 : 0);
        started[t] = 0;
    }
    for (int t = 1; t < nt; ++t) started[t] = pthread_create(&th[t], NULL, list_worker, &jobs[t]) == 0;
    list_worker(&jobs[0]);
    Py_ssize_t big = jobs[0].big;
    for (int t = 1; t < nt; ++t) {
        if (started[t]) pthread_join(th[t], NULL); else list_worker(&jobs[t]);
        big += jobs[t].big;
    }
    /* the list now refers to the cached ints tally[v] more times: their reference counts follow (this thread holds the GIL) */
    for (int v = 0; v < NSMALL; ++v) {
        Py_ssize_t c = 0;
        for (int t = 0; t < nt; ++t) c += jobs[t].tally[v];
        if (c) Py_SET_REFCNT(g_small[v], Py_REFCNT(g_small[v]) + c);
    }
    free(jobs);
    if (big) {
        const uint32_t* src = (const uint32_t*)view.buf;
        for (Py_ssize_t i = 0; i < n; ++i)
            if (!dst[i]) {
                PyObject* o = PyLong_FromUnsignedLong(src[i]);
                if (!o) {                                   /* out of memory: slots still NULL are legal for list deallocation */
                    PyBuffer_Release(&view);
                    Py_DECREF(list);
                    return NULL;
                }
                dst[i] = o;
            }
    }
    PyBuffer_Release(&view);
    return list;
}

static PyMethodDef methods[] = {
    {"join_fixed", join_fixed, METH_VARARGS, "join_fixed(seq, k[, threads]) -> bytes of len(seq)*k, or None if an item is not a k-character ASCII str / bytes"},
    {"join_fixed_into", join_fixed_into, METH_VARARGS, "join_fixed_into(seq, k, buffer[, threads]) -> True, or None if an item is not a k-character ASCII str / bytes"},
    {"ascii_view", ascii_view, METH_O, "ascii_view(s) -> read-only memoryview over the characters of an ASCII str, or None"},
    {"u32_list", u32_list, METH_VARARGS, "u32_list(buffer, n[, threads]) -> list[int] of the first n uint32 of the buffer"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_aix_pyfast", "list[str] / list[int] conversions for aindex_amd", -1, methods, NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__aix_pyfast(void) {
    PyObject* m = PyModule_Create(&moddef);
    if (!m) return NULL;
    for (int v = 0; v < NSMALL; ++v) {
        g_small[v] = PyLong_FromLong(v);                  /* the cached object (CPython caches -5 .. 256) */
        if (!g_small[v]) { Py_DECREF(m); return NULL; }
    }
    return m;
}
