// Test program for include/tf_compat/python2.7/Python.h: the CPython call sequence the reference's HM side makes to read
// the training mean (hevc/hm_common/c++/source_common/interface_c_python.cpp + TComPrediction.cpp:181-236 are the model
// for the sequence; this is not a copy of them).  usage: py_shadow_sample <module> <function> <file>
// Prints the float with 17 significant digits, or the error the shadow interpreter reports, and exits 0 / 1.
#include "python2.7/Python.h"

int main(int argc, char** argv)
{
    if (argc != 4) return 64;
    Py_Initialize();
    if (!Py_IsInitialized()) return 2;
    char name[] = "path";
    PyObject* sys_path = PySys_GetObject(name);                  // borrowed
    PyObject* dir = PyString_FromString(".");
    if (!sys_path || !dir || PyList_Insert(sys_path, 0, dir) < 0) return 3;
    Py_DECREF(dir);
    PyObject* mod_name = PyString_FromString(argv[1]);
    PyObject* mod = PyImport_Import(mod_name);
    Py_DECREF(mod_name);
    if (!mod) { if (PyErr_Occurred()) PyErr_Print(); return 1; }
    PyObject* fn = PyObject_GetAttrString(mod, argv[2]);
    Py_DECREF(mod);
    if (!fn) { if (PyErr_Occurred()) PyErr_Print(); return 1; }
    if (!PyCallable_Check(fn)) return 4;
    PyObject* arg = PyString_FromString(argv[3]);
    PyObject* res = PyObject_CallFunctionObjArgs(fn, arg, NULL);
    Py_DECREF(arg);
    Py_DECREF(fn);
    if (!res) { if (PyErr_Occurred()) PyErr_Print(); return 1; }
    if (!PyFloat_CheckExact(res)) return 5;
    printf("%.17g\n", PyFloat_AsDouble(res));
    Py_DECREF(res);
    if (PyErr_Occurred()) return 6;
    Py_Finalize();
    return Py_IsInitialized() ? 7 : 0;
}
