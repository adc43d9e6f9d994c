"""The reference-shaped C APIs on a CPU-only box: argument checks happen before any device work, in the reference's
order, with the reference's codes; builders behave like the reference's opaque Box'ed builders.

Mirrors the reference's own C-API tests: core  bc1 c_api/transform_with_settings.rs:155-260, c_api/transform_auto.rs:
233-; stable  bc1-api c_api/transform/manual_transform_builder.rs tests, auto_transform_builder.rs tests, error.rs tests."""
import ctypes as C

import numpy as np
import pytest

import cabi


@pytest.fixture(scope="module")
def lib(pkg):
    return cabi.bind(C.CDLL(pkg._lib.lib_path()))


DATA16 = bytes([0x00, 0x01, 0x02, 0x03, 0x80, 0x81, 0x82, 0x83, 0x04, 0x05, 0x06, 0x07, 0x84, 0x85, 0x86, 0x87])


def buf(n):
    return np.zeros(n, dtype=np.uint8)


@pytest.mark.parametrize("n,S", [(1, cabi.CoreSettings2), (2, cabi.CoreSettings2), (3, cabi.CoreSettings3)])
def test_core_null_and_length_checks(lib, n, S):
    # core codes: NullDataPointer 1, NullOutputBufferPointer 2, InvalidDataLength 5, OutputBufferTooSmall 6
    x, y = np.frombuffer(DATA16 * 2, dtype=np.uint8).copy(), buf(32)
    s = S()
    s.DecorrelationMode = 1
    s.SplitColourEndpoints = True
    for d in ("transform", "untransform"):
        f = getattr(lib, f"dltbc{n}core_{d}")
        assert f(None, 32, y.ctypes.data, 32, s).ErrorCode == 1
        assert f(x.ctypes.data, 32, None, 32, s).ErrorCode == 2
        assert f(None, 32, None, 32, s).ErrorCode == 1  # input is checked first
        assert f(x.ctypes.data, 15, y.ctypes.data, 32, s).ErrorCode == 5
        assert f(x.ctypes.data, 32, y.ctypes.data, 31, s).ErrorCode == 6
        assert f(x.ctypes.data, 15, y.ctypes.data, 1, s).ErrorCode == 5  # length before size


@pytest.mark.parametrize("n,S", [(1, cabi.CoreSettings2), (2, cabi.CoreSettings2), (3, cabi.CoreSettings3)])
def test_core_auto_null_checks(lib, n, S):
    x, y = np.frombuffer(DATA16 * 2, dtype=np.uint8).copy(), buf(32)
    est, _ = cabi.make_estimator("dummy")
    out = S()
    f = getattr(lib, f"dltbc{n}core_transform_auto")
    a = cabi.AutoSettings(False)
    assert f(None, 32, y.ctypes.data, 32, C.byref(est), a, C.byref(out)).ErrorCode == 1
    assert f(x.ctypes.data, 32, None, 32, C.byref(est), a, C.byref(out)).ErrorCode == 2
    assert f(x.ctypes.data, 32, y.ctypes.data, 32, None, a, C.byref(out)).ErrorCode == 3
    assert f(x.ctypes.data, 32, y.ctypes.data, 32, C.byref(est), a, None).ErrorCode == 4
    assert f(x.ctypes.data, 15, y.ctypes.data, 32, C.byref(est), a, C.byref(out)).ErrorCode == 5
    assert f(x.ctypes.data, 32, y.ctypes.data, 16, C.byref(est), a, C.byref(out)).ErrorCode == 6


@pytest.mark.parametrize("n", [1, 2, 3])
def test_stable_manual_builder_lifecycle_and_checks(lib, n):
    p = f"dltbc{n}_"
    b = getattr(lib, p + "new_ManualTransformBuilder")()
    assert b
    getattr(lib, p + "free_ManualTransformBuilder")(None)  # NULL is fine
    assert getattr(lib, p + "clone_ManualTransformBuilder")(None) is None
    c = getattr(lib, p + "clone_ManualTransformBuilder")(b)
    assert c and c != b
    # setters ignore NULL
    getattr(lib, p + "ManualTransformBuilder_SetDecorrelationMode")(None, 0)
    getattr(lib, p + "ManualTransformBuilder_SetSplitColourEndpoints")(None, True)
    getattr(lib, p + "ManualTransformBuilder_ResetToDefaults")(None)
    x, y = np.frombuffer(DATA16 * 2, dtype=np.uint8).copy(), buf(32)
    for d in ("Transform", "Untransform"):
        f = getattr(lib, p + "ManualTransformBuilder_" + d)
        # stable codes: NullDataPointer 5, NullOutputBufferPointer 9, NullManualTransformBuilderPointer 10,
        # InvalidLength 1, OutputBufferTooSmall 2
        assert f(None, 32, y.ctypes.data, 32, b).ErrorCode == 5
        assert f(x.ctypes.data, 32, None, 32, b).ErrorCode == 9
        assert f(x.ctypes.data, 32, y.ctypes.data, 32, None).ErrorCode == 10
        assert f(None, 32, None, 32, None).ErrorCode == 5
        assert f(x.ctypes.data, 17, y.ctypes.data, 32, b).ErrorCode == 1
        assert f(x.ctypes.data, 32, y.ctypes.data, 8, b).ErrorCode == 2
    getattr(lib, p + "free_ManualTransformBuilder")(b)
    getattr(lib, p + "free_ManualTransformBuilder")(c)


@pytest.mark.parametrize("n", [1, 2, 3])
def test_stable_auto_builder_checks(lib, n):
    p = f"dltbc{n}_"
    assert getattr(lib, p + "new_AutoTransformBuilder")(None) is None
    est, _ = cabi.make_estimator("dummy")
    b = getattr(lib, p + "new_AutoTransformBuilder")(C.byref(est))
    assert b
    assert getattr(lib, p + "AutoTransformBuilder_SetUseAllDecorrelationModes")(None, True).ErrorCode == 11
    assert getattr(lib, p + "AutoTransformBuilder_SetUseAllDecorrelationModes")(b, True).ErrorCode == 0
    x, y = np.frombuffer(DATA16 * 2, dtype=np.uint8).copy(), buf(32)
    out = C.c_void_p(0x1234)
    f = getattr(lib, p + "AutoTransformBuilder_Transform")
    assert f(None, x.ctypes.data, 32, y.ctypes.data, 32, C.byref(out)).ErrorCode == 11
    assert f(b, None, 32, y.ctypes.data, 32, C.byref(out)).ErrorCode == 5
    assert f(b, x.ctypes.data, 32, None, 32, C.byref(out)).ErrorCode == 9
    assert f(b, x.ctypes.data, 32, y.ctypes.data, 32, None).ErrorCode == 12
    assert f(b, x.ctypes.data, 17, y.ctypes.data, 32, C.byref(out)).ErrorCode == 1
    assert out.value is None  # set to NULL on failure
    out = C.c_void_p(0x1234)
    assert f(b, x.ctypes.data, 32, y.ctypes.data, 16, C.byref(out)).ErrorCode == 2
    assert out.value is None
    getattr(lib, p + "free_AutoTransformBuilder")(b)
    getattr(lib, p + "free_AutoTransformBuilder")(None)


def test_error_messages(lib):
    # bc1-api c_api/error.rs:131-175 and its tests
    assert lib.dltbc1_error_message(0) == b"Success"
    assert lib.dltbc1_error_message(1) == b"Invalid input length: Length must be divisible by 8 (BC1 block size)"
    assert lib.dltbc2_error_message(1) == b"Invalid input length: Length must be divisible by 16 (BC2 block size)"
    assert lib.dltbc1_error_message(2) == b"Output buffer too small for the operation"
    assert lib.dltbc1_error_message(3) == b"Memory allocation failed"
    assert lib.dltbc1_error_message(10) == b"Null pointer provided for Dltbc1ManualTransformBuilder parameter"
    assert lib.dltbc2_error_message(12) == b"Null pointer provided for manual builder output parameter"
    for code in range(13):
        assert lib.dltbc1_error_message(code) and lib.dltbc2_error_message(code)


@pytest.mark.parametrize("n", [1, 2, 3])
def test_stable_api_tells_a_missing_device_from_an_allocation_failure(lib, n):
    """Additive codes above the reference's range: on a box without a usable GPU a valid call reports DeviceUnavailable
    (100), not the reference's AllocationFailed (3); both have a message."""
    torch = pytest.importorskip("torch")
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    p = f"dltbc{n}_"
    b = getattr(lib, p + "new_ManualTransformBuilder")()
    x, y = np.frombuffer(DATA16 * 2, dtype=np.uint8).copy(), buf(32)
    assert getattr(lib, p + "ManualTransformBuilder_Transform")(x.ctypes.data, 32, y.ctypes.data, 32, b).ErrorCode == 100
    assert getattr(lib, p + "ManualTransformBuilder_Transform")(x.ctypes.data, 0, y.ctypes.data, 0, b).ErrorCode == 0   # nothing to do
    getattr(lib, p + "free_ManualTransformBuilder")(b)
    msg = getattr(lib, p + "error_message")
    assert b"HIP device" in msg(100) and b"HIP runtime" in msg(101) and msg(3) == b"Memory allocation failed"


def test_struct_layouts_match_headers():
    # core: {bool, u8} = 2 bytes; BC3 additive {bool, bool, u8} = 3 bytes; estimator = 3 pointers
    assert C.sizeof(cabi.CoreSettings2) == 2 and C.sizeof(cabi.CoreSettings3) == 3
    assert C.sizeof(cabi.DltSizeEstimator) == 3 * C.sizeof(C.c_void_p)
    assert C.sizeof(cabi.Result) == 4


class _StableResult(C.Structure):
    _fields_ = [("ErrorCode", C.c_int32)]


def _bind_bc7(pkg):
    l = C.CDLL(pkg._lib.lib_path())
    vp, sz = C.c_void_p, C.c_size_t
    l.dltbc7_new_ManualTransformBuilder.restype = vp
    l.dltbc7_free_ManualTransformBuilder.argtypes = [vp]
    l.dltbc7_clone_ManualTransformBuilder.argtypes, l.dltbc7_clone_ManualTransformBuilder.restype = [vp], vp
    l.dltbc7_ManualTransformBuilder_ResetToDefaults.argtypes = [vp]
    for d in ("Transform", "Untransform"):
        f = getattr(l, "dltbc7_ManualTransformBuilder_" + d)
        f.argtypes, f.restype = [vp, sz, vp, sz, vp], _StableResult
    l.dltbc7_error_message.argtypes, l.dltbc7_error_message.restype = [C.c_int32], C.c_char_p
    return l


def test_bc7_stable_style_builder_checks(pkg):
    """include/dltbc7.h (additive): the BC1/BC2 builder shape, codes and check order for this build's BC7 format."""
    l = _bind_bc7(pkg)
    b = l.dltbc7_new_ManualTransformBuilder()
    assert b
    l.dltbc7_free_ManualTransformBuilder(None)
    assert l.dltbc7_clone_ManualTransformBuilder(None) is None
    c = l.dltbc7_clone_ManualTransformBuilder(b)
    assert c and c != b
    l.dltbc7_ManualTransformBuilder_ResetToDefaults(None)
    x, y = np.frombuffer(DATA16 * 2, dtype=np.uint8).copy(), buf(32)
    for d in ("Transform", "Untransform"):
        f = getattr(l, "dltbc7_ManualTransformBuilder_" + d)
        assert f(None, 32, y.ctypes.data, 32, b).ErrorCode == 5
        assert f(x.ctypes.data, 32, None, 32, b).ErrorCode == 9
        assert f(x.ctypes.data, 32, y.ctypes.data, 32, None).ErrorCode == 10
        assert f(x.ctypes.data, 17, y.ctypes.data, 32, b).ErrorCode == 1
        assert f(x.ctypes.data, 32, y.ctypes.data, 8, b).ErrorCode == 2
        assert f(x.ctypes.data, 0, y.ctypes.data, 0, b).ErrorCode == 0      # nothing to do: no device needed
    assert l.dltbc7_error_message(1) == b"Invalid input length: Length must be divisible by 16 (BC7 block size)"
    assert l.dltbc7_error_message(10) == b"Null pointer provided for Dltbc7ManualTransformBuilder parameter"
    l.dltbc7_free_ManualTransformBuilder(b)
    l.dltbc7_free_ManualTransformBuilder(c)


@pytest.mark.gpu
def test_bc7_stable_style_builder_round_trip(pkg, oracle):
    l = _bind_bc7(pkg)
    b = l.dltbc7_new_ManualTransformBuilder()
    x = np.fromfile(__import__("os").path.join(__import__("helpers").GOLDEN, "r2-256-bc7.payload.bin"), dtype=np.uint8)
    y, z = np.zeros_like(x), np.zeros_like(x)
    assert l.dltbc7_ManualTransformBuilder_Transform(x.ctypes.data, x.size, y.ctypes.data, y.size, b).ErrorCode == 0
    assert np.array_equal(y, oracle.transform_bc7(x))
    assert l.dltbc7_ManualTransformBuilder_Untransform(y.ctypes.data, y.size, z.ctypes.data, z.size, b).ErrorCode == 0
    assert np.array_equal(z, x)
    l.dltbc7_free_ManualTransformBuilder(b)
