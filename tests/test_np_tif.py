"""np_tif mirror: round trips (the reference's own self-test, np_tif.py:445-462,
re-expressed), ImageJ layout, and -- in the build container only -- reading the
reference's real ImageJ-written test objects."""
import os

import numpy as np
import pytest

from rescan_line_sted_amd import np_tif

REF = '/root/reference/figure_generation'


@pytest.mark.parametrize('shape', [(7,), (5, 6), (3, 5, 6), (1, 128, 128)])
@pytest.mark.parametrize('dtype', ['uint8', 'uint16', 'uint32', 'int16', 'int32', 'float32', 'float64', 'int64', 'uint64'])
def test_round_trip(tmp_path, shape, dtype):
    rng = np.random.default_rng(1)
    a = (rng.random(shape) * 200).astype(dtype)
    fn = str(tmp_path / 'a.tif')
    np_tif.array_to_tif(a, fn)
    b = np_tif.tif_to_array(fn)
    want = a.reshape((1,) * (3 - a.ndim) + a.shape)
    assert b.shape == want.shape
    coerced = {'float64': 'float32', 'int64': 'int32', 'uint64': 'uint32'}.get(dtype, dtype)
    assert b.dtype == np.dtype(coerced)                      # 64 -> 32 bit coercion, np_tif.py:145-151
    assert np.array_equal(b, want.astype(coerced))
    np_tif.array_to_tif(a, fn, coerce_64bit_to_32bit=False)
    c = np_tif.tif_to_array(fn)
    assert c.dtype == np.dtype(dtype) and np.array_equal(c, want)


def test_imagej_layout_and_description(tmp_path):
    a = np.arange(2 * 3 * 4 * 5, dtype=np.float32).reshape(6, 4, 5)
    fn = str(tmp_path / 'h.tif')
    np_tif.array_to_tif(a, fn, slices=3, channels=2)
    data, desc = np_tif.tif_to_array(fn, image_descriptions=True)
    assert np.array_equal(data, a)
    assert len(desc) == 1 and 'images=6' in desc[0] and 'channels=2' in desc[0] and 'slices=3' in desc[0]
    ifds, endian = np_tif.parse_tif(fn)
    assert endian == 'little' and len(ifds) == 6
    # header, first IFD, description, then all pixel data contiguous
    offs = [d['StripOffsets'] for d in ifds]
    assert offs == [offs[0] + i * 4 * 5 * 4 for i in range(6)]
    with pytest.raises(AssertionError):
        np_tif.array_to_tif(a, fn, slices=4, channels=2)


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree only exists in the build container')
def test_reads_the_reference_test_objects(golden):
    objs = golden('objects')
    for name in ('cat', 'astronaut', 'lines', 'rings'):
        a = np_tif.tif_to_array(os.path.join(REF, 'test_object_%s.tif' % name))
        assert a.dtype == objs[name].dtype and np.array_equal(a, objs[name])
