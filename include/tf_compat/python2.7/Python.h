// python2.7/Python.h -- include-path shadow of the CPython header for the reference's modified HM-16.15.
//
// The reference embeds CPython 2.7 for ONE purpose: to unpickle one float, the training-set mean
// (hevc/hm_common/c++/source_common/interface_c_python.{h,cpp}; call sequence at
// hevc/hm_16_15_substitution/source/Lib/TLibCommon/TComPrediction.cpp:181-236, same lines in hm_16_15_switch;
// the Python side is the four-line hevc/hm_common/loading.py: `pickle.load(open(path, 'rb'))`).
// This header gives those two files the 17 CPython names they use, backed by a parser for a pickled float, so that
// both HM variants compile unchanged and link with neither libpython nor TensorFlow (plus the seven more names the
// reference's own test program uses -- hevc/hm_common/c++/source_test/tests.cpp:3-90,823-892: PyList_Check / _Size /
// _GetItem, PyString_AsString, PyInt_CheckExact / PyInt_AsLong for its pickled integer; tools/hm builds that program too):
//
//   Py_Initialize / Py_IsInitialized / Py_Finalize
//   PySys_GetObject("path"), PyString_FromString, PyList_Insert, Py_DECREF          (append_sys_path)
//   PyImport_Import("loading"), PyObject_GetAttrString(m, "load_via_pickle"), PyCallable_Check   (get_callable)
//   PyObject_CallFunctionObjArgs(f, path, NULL)                                      (load_via_pickle)
//   PyFloat_CheckExact, PyFloat_AsDouble, PyErr_Occurred, PyErr_Print
//
// Behaviour kept: a missing module / attribute / file sets an error and returns NULL (the callers print it with
// PyErr_Print and assert); `load_via_pickle` accepts pickle protocols 0-4 of a float (the reference's file is
// protocol 2: 80 02 'G' <8-byte big-endian double> '.') and, as an extension, a text file holding the number.
// Not an interpreter: the only module is `loading`, the only callable `load_via_pickle`.
#ifndef PNN_PYTHON_SHADOW_H
#define PNN_PYTHON_SHADOW_H

// "Python.h" implies these (interface_c_python.h:4-8 relies on it).
#include <assert.h>
#include <errno.h>
#include <limits.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#define PY_MAJOR_VERSION 2
#define PY_MINOR_VERSION 7
#define PY_VERSION "2.7-pnn-shadow"

struct _object {
    enum Kind { STRING, LIST, MODULE, FUNCTION, FLOAT, INT, ERROR_STATE };
    long ob_refcnt;
    Kind kind;
    std::string s;                      // STRING: value; MODULE / FUNCTION: name; ERROR_STATE: message
    std::vector<_object*> items;        // LIST
    double d;                           // FLOAT
    long i;                             // INT
    explicit _object(Kind k) : ob_refcnt(1), kind(k), d(0.), i(0) {}
};
typedef struct _object PyObject;

namespace pnn_py {

struct Interp {
    bool initialized;
    PyObject* sys_path;
    PyObject* error;                    // pending exception (NULL = none)
    Interp() : initialized(false), sys_path(NULL), error(NULL) {}
};
inline Interp& interp() { static Interp i; return i; }

inline void release(PyObject* o)
{
    if (!o || --o->ob_refcnt > 0) return;
    for (size_t i = 0; i < o->items.size(); i++) release(o->items[i]);
    delete o;
}

inline PyObject* set_error(const std::string& msg)
{
    Interp& in = interp();
    release(in.error);
    in.error = new _object(_object::ERROR_STATE);
    in.error->s = msg;
    return NULL;
}

// pickle.load of a float or an int: protocol 0 ('F' / 'I' + repr + '\n'), protocols 1-4 ('G' + big-endian IEEE double; 'K' /
// 'M' / 'J' = 1- / 2- / 4-byte little-endian int, 0x8a = LONG1), with the PROTO / FRAME / MEMOIZE / PUT framing opcodes such a
// pickle can carry; anything else is "not a number pickle".  *is_int tells which of *out / *iout holds the value.
inline bool unpickle_number(const std::vector<unsigned char>& b, double* out, long* iout, bool* is_int)
{
    size_t i = 0;
    *is_int = false;
    auto ends = [&](size_t j) { return j < b.size() && (b[j] == '.' || b[j] == 'q' || b[j] == 0x94 || b[j] == 'p'); };
    while (i < b.size()) {
        const unsigned char op = b[i++];
        switch (op) {
        case 0x80: i += 1; break;                                        // PROTO n
        case 0x95: i += 8; break;                                        // FRAME len8
        case 0x94: break;                                                // MEMOIZE
        case 'q': i += 1; break;                                         // BINPUT
        case 'r': i += 4; break;                                         // LONG_BINPUT
        case 'p': while (i < b.size() && b[i] != '\n') i++; i++; break;  // PUT
        case 'G': {
            if (i + 8 > b.size()) return false;
            unsigned long long u = 0;
            for (int k = 0; k < 8; k++) u = (u << 8) | b[i + k];
            memcpy(out, &u, 8);
            return ends(i + 8);
        }
        case 'F': {
            std::string t;
            while (i < b.size() && b[i] != '\n') t.push_back((char)b[i++]);
            char* end = NULL;
            *out = strtod(t.c_str(), &end);
            return end != t.c_str();
        }
        case 'K': case 'M': case 'J': case 0x8a: {
            size_t n = op == 'K' ? 1 : op == 'M' ? 2 : 4;
            if (op == 0x8a) { if (i >= b.size()) return false; n = b[i++]; }
            if (n > 8 || i + n > b.size()) return false;
            unsigned long long u = 0;
            for (size_t k = 0; k < n; k++) u |= (unsigned long long)b[i + k] << (8 * k);
            if ((op == 'J' || op == 0x8a) && n > 0 && n < 8 && (b[i + n - 1] & 0x80)) u |= ~0ull << (8 * n);   // sign extension
            *iout = (long)u; *is_int = true;
            return ends(i + n);
        }
        case 'I': case 'L': {
            std::string t;
            while (i < b.size() && b[i] != '\n') t.push_back((char)b[i++]);
            char* end = NULL;
            *iout = strtol(t.c_str(), &end, 10); *is_int = true;
            return end != t.c_str();
        }
        default: return false;
        }
    }
    return false;
}

inline PyObject* load_via_pickle(const std::string& path)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return set_error("IOError: [Errno 2] No such file or directory: '" + path + "'");
    std::vector<unsigned char> bytes;
    unsigned char buf[256];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) bytes.insert(bytes.end(), buf, buf + n);
    fclose(f);
    double v = 0.;
    long iv = 0;
    bool is_int = false;
    if (!unpickle_number(bytes, &v, &iv, &is_int)) {
        // extension: the mean as text ("117.8952234192841")
        const std::string t(bytes.begin(), bytes.end());
        char* end = NULL;
        v = strtod(t.c_str(), &end);
        while (end && *end && strchr(" \t\r\n", *end)) end++;
        if (bytes.empty() || end == t.c_str() || (end && *end))
            return set_error("UnpicklingError: '" + path + "' holds neither a pickled float nor a number");
    }
    PyObject* o = new _object(is_int ? _object::INT : _object::FLOAT);
    o->d = v; o->i = iv;
    return o;
}

}  // namespace pnn_py

#define Py_INCREF(o) (++(o)->ob_refcnt)
#define Py_DECREF(o) ::pnn_py::release(o)
#define Py_XDECREF(o) ::pnn_py::release(o)

inline void Py_Initialize(void)
{
    pnn_py::Interp& in = pnn_py::interp();
    if (in.initialized) return;
    in.sys_path = new _object(_object::LIST);
    in.initialized = true;
}
inline int Py_IsInitialized(void) { return pnn_py::interp().initialized ? 1 : 0; }
inline void Py_Finalize(void)
{
    pnn_py::Interp& in = pnn_py::interp();
    pnn_py::release(in.sys_path); in.sys_path = NULL;
    pnn_py::release(in.error); in.error = NULL;
    in.initialized = false;
}

inline PyObject* PyErr_Occurred(void) { return pnn_py::interp().error; }      // borrowed
inline void PyErr_Print(void)
{
    pnn_py::Interp& in = pnn_py::interp();
    if (!in.error) return;
    fprintf(stderr, "%s\n", in.error->s.c_str());
    pnn_py::release(in.error);
    in.error = NULL;
}
inline void PyErr_Clear(void) { pnn_py::Interp& in = pnn_py::interp(); pnn_py::release(in.error); in.error = NULL; }

inline PyObject* PySys_GetObject(char* name)                                   // borrowed; NULL without an exception
{
    pnn_py::Interp& in = pnn_py::interp();
    return (in.initialized && name && !strcmp(name, "path")) ? in.sys_path : NULL;
}

inline PyObject* PyString_FromString(const char* v)
{
    if (!v) return pnn_py::set_error("SystemError: NULL string");
    PyObject* o = new _object(_object::STRING);
    o->s = v;
    return o;
}
inline PyObject* PyUnicode_FromString(const char* v) { return PyString_FromString(v); }

inline int PyList_Insert(PyObject* list, long where, PyObject* item)
{
    if (!list || list->kind != _object::LIST || !item) { pnn_py::set_error("SystemError: bad argument to PyList_Insert"); return -1; }
    long n = (long)list->items.size();
    if (where < 0) { where += n; if (where < 0) where = 0; }
    if (where > n) where = n;
    Py_INCREF(item);
    list->items.insert(list->items.begin() + where, item);
    return 0;
}

inline PyObject* PyImport_Import(PyObject* name)
{
    if (!name || name->kind != _object::STRING) return pnn_py::set_error("TypeError: module name must be a string");
    if (name->s != "loading") return pnn_py::set_error("ImportError: No module named " + name->s);
    PyObject* m = new _object(_object::MODULE);
    m->s = name->s;
    return m;
}

inline PyObject* PyObject_GetAttrString(PyObject* o, const char* attr)
{
    if (!o || !attr) return pnn_py::set_error("SystemError: NULL argument to PyObject_GetAttrString");
    if (o->kind == _object::MODULE && o->s == "loading" && !strcmp(attr, "load_via_pickle")) {
        PyObject* f = new _object(_object::FUNCTION);
        f->s = attr;
        return f;
    }
    return pnn_py::set_error(std::string("AttributeError: object has no attribute '") + attr + "'");
}

inline int PyCallable_Check(PyObject* o) { return o && o->kind == _object::FUNCTION; }

// Calls `callable(arg0, ...)`; the argument list ends with NULL. Only `loading.load_via_pickle(path)` exists.
inline PyObject* PyObject_CallFunctionObjArgs(PyObject* callable, ...)
{
    va_list ap;
    va_start(ap, callable);
    PyObject* arg0 = va_arg(ap, PyObject*);
    PyObject* arg1 = arg0 ? va_arg(ap, PyObject*) : NULL;
    va_end(ap);
    if (!PyCallable_Check(callable)) return pnn_py::set_error("TypeError: object is not callable");
    if (!arg0 || arg1 || arg0->kind != _object::STRING)
        return pnn_py::set_error("TypeError: load_via_pickle() takes exactly 1 argument (a path)");
    return pnn_py::load_via_pickle(arg0->s);
}

typedef long Py_ssize_t;
inline int PyList_Check(PyObject* o) { return o && o->kind == _object::LIST; }
inline Py_ssize_t PyList_Size(PyObject* o) { return PyList_Check(o) ? (Py_ssize_t)o->items.size() : -1; }
inline PyObject* PyList_GetItem(PyObject* o, Py_ssize_t i)                      // borrowed
{
    if (!PyList_Check(o) || i < 0 || i >= (Py_ssize_t)o->items.size()) return pnn_py::set_error("IndexError: list index out of range");
    return o->items[(size_t)i];
}
inline char* PyString_AsString(PyObject* o)
{
    if (!o || o->kind != _object::STRING) { pnn_py::set_error("TypeError: expected string"); return NULL; }
    return const_cast<char*>(o->s.c_str());
}
inline int PyInt_CheckExact(PyObject* o) { return o && o->kind == _object::INT; }
inline int PyLong_CheckExact(PyObject* o) { return PyInt_CheckExact(o); }
inline long PyInt_AsLong(PyObject* o)
{
    if (!PyInt_CheckExact(o)) { pnn_py::set_error("TypeError: an integer is required"); return -1; }
    return o->i;
}
inline long PyLong_AsLong(PyObject* o) { return PyInt_AsLong(o); }

inline int PyFloat_CheckExact(PyObject* o) { return o && o->kind == _object::FLOAT; }
inline int PyFloat_Check(PyObject* o) { return PyFloat_CheckExact(o); }
inline double PyFloat_AsDouble(PyObject* o)
{
    if (!PyFloat_CheckExact(o)) { pnn_py::set_error("TypeError: a float is required"); return -1.0; }
    return o->d;
}

#endif  // PNN_PYTHON_SHADOW_H
