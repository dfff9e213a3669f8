import ctypes, numpy as np
L = ctypes.CDLL("knn-for-homology_amd/libknn355.so")
L.knn_last_error.restype = ctypes.c_char_p
L.knn_flat_create.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p)]
L.knn_flat_add.argtypes    = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
L.knn_flat_search.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                              ctypes.c_void_p, ctypes.c_void_p]
L.knn_normalize_l2.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32]
L.knn_free.argtypes = [ctypes.c_void_p]

def check(rc):
    if rc: raise RuntimeError(L.knn_last_error().decode())

x = np.random.default_rng(0).standard_normal((3000, 1024)).astype(np.float32)
check(L.knn_normalize_l2(x.ctypes.data, *x.shape))
h = ctypes.c_void_p()
check(L.knn_flat_create(x.shape[1], 0, ctypes.byref(h)))
check(L.knn_flat_add(h, x.ctypes.data, x.shape[0]))
D = np.empty((x.shape[0], 11), np.float32); I = np.empty((x.shape[0], 11), np.int64)
check(L.knn_flat_search(h, x.ctypes.data, x.shape[0], 11, D.ctypes.data, I.ctypes.data))
L.knn_free(h)
assert (I[:, 0] == np.arange(3000)).all() and np.allclose(D[:, 0], 1.0, atol=1e-5)
print("integration snippet ok", D[0, :3], I[0, :3])
