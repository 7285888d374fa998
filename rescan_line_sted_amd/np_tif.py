"""Minimal ImageJ-style TIFF reader/writer with the call surface of the
reference's figure_generation/np_tif.py (tif_to_array :13-112, array_to_tif
:114-198, parse_tif :200-...).  Host I/O only; written from the TIFF 6.0
layout with `struct`, not derived from the reference's implementation.

Scope (same as the reference): uncompressed, one sample per pixel, 8/16/32/64
bit unsigned / signed / float stacks; a stack is a chain of IFDs of identical
geometry.  Files are written the way ImageJ lays them out -- 8-byte header,
first IFD, image description, all pixel data, remaining IFDs -- so ImageJ opens
them as stacks / hyperstacks.
"""
import struct

import numpy as np

_TAG_NAMES = {
    254: 'NewSubFileType', 256: 'ImageWidth', 257: 'ImageLength', 258: 'BitsPerSample',
    259: 'Compression', 262: 'PhotometricInterpretation', 270: 'ImageDescription',
    273: 'StripOffsets', 277: 'SamplesPerPixel', 278: 'RowsPerStrip', 279: 'StripByteCounts',
    282: 'XResolution', 283: 'YResolution', 296: 'ResolutionUnit', 339: 'SampleFormat',
}
# TIFF field type -> (struct code, byte size)
_FIELD = {1: ('B', 1), 2: ('c', 1), 3: ('H', 2), 4: ('I', 4), 5: ('II', 8), 6: ('b', 1), 7: ('B', 1),
          8: ('h', 2), 9: ('i', 4), 10: ('ii', 8), 11: ('f', 4), 12: ('d', 8), 16: ('Q', 8)}
_SAMPLE_FORMAT = {'u': 1, 'i': 2, 'f': 3}
_KIND = {1: 'uint', 2: 'int', 3: 'float'}


def _read_value(f, order, ftype, count, value_field):
    code, size = _FIELD[ftype]
    nbytes = size * count
    if nbytes <= 4:
        raw = value_field[:nbytes]
    else:
        (offset,) = struct.unpack(order + 'I', value_field)
        here = f.tell()
        f.seek(offset)
        raw = f.read(nbytes)
        f.seek(here)
    if ftype == 2:
        return raw.split(b'\x00')[0].decode('ascii', errors='replace')
    vals = struct.unpack(order + code * count, raw)
    if ftype in (5, 10):
        vals = tuple(vals[i] / vals[i + 1] if vals[i + 1] else 0.0 for i in range(0, len(vals), 2))
    return vals[0] if len(vals) == 1 else vals


def parse_tif(filename, verbose=False):
    """Returns (ifds, endian): a list of {tag name: value} dicts, one per image
    file directory, and 'little' or 'big'."""
    with open(filename, 'rb') as f:
        head = f.read(8)
        if head[:2] == b'II':
            order, endian = '<', 'little'
        elif head[:2] == b'MM':
            order, endian = '>', 'big'
        else:
            raise UserWarning("Not a TIF file")
        magic, offset = struct.unpack(order + 'HI', head[2:])
        if magic != 42:
            raise UserWarning("Not a TIF file")
        ifds = []
        while offset != 0:
            f.seek(offset)
            (n_entries,) = struct.unpack(order + 'H', f.read(2))
            entries = f.read(12 * n_entries)
            (offset,) = struct.unpack(order + 'I', f.read(4))
            ifd = {}
            for e in range(n_entries):
                tag, ftype, count = struct.unpack(order + 'HHI', entries[12 * e:12 * e + 8])
                if ftype not in _FIELD:
                    continue
                value = _read_value(f, order, ftype, count, entries[12 * e + 8:12 * e + 12])
                ifd[_TAG_NAMES.get(tag, tag)] = value
            if verbose:
                print("IFD:", ifd)
            ifds.append(ifd)
    return ifds, endian


def tif_to_array(filename, image_descriptions=False, verbose=False):
    """Load a TIF stack as a numpy array of shape (n_images, length, width)."""
    ifds, endian = parse_tif(filename, verbose)
    first = ifds[0]
    geometry = ('ImageWidth', 'ImageLength', 'BitsPerSample')
    for d in ifds:
        if (any(d[k] != first[k] for k in geometry) or
                d.get('SampleFormat', 1) != first.get('SampleFormat', 1)):
            raise UserWarning("The TIF we're trying to load has mismatched IFD's")
        if (d.get('SamplesPerPixel', 1) != 1 or d.get('NewSubFileType', 0) != 0 or
                d.get('Compression', 1) != 1 or d.get('PhotometricInterpretation', 0) not in (0, 1)):
            raise UserWarning("The TIF we're trying to load uses options that np_tif doesn't support.")
    width, length, bits = (int(first[k]) for k in geometry)
    kind = _KIND.get(int(first.get('SampleFormat', 1)))
    try:
        dtype = np.dtype(getattr(np, '%s%d' % (kind, bits)))
    except (AttributeError, TypeError):
        raise UserWarning("Unsupported data format: %s%d" % (kind, bits))
    chunks = []
    with open(filename, 'rb') as f:
        for d in ifds:
            offsets, counts = d['StripOffsets'], d['StripByteCounts']
            if not isinstance(offsets, tuple):
                offsets, counts = (offsets,), (counts,)
            for off, cnt in zip(offsets, counts):
                f.seek(int(off))
                chunks.append(f.read(int(cnt)))
    data = np.frombuffer(b''.join(chunks), dtype=dtype.newbyteorder('<' if endian == 'little' else '>'))
    data = data.astype(dtype).reshape(len(ifds), length, width)
    if image_descriptions:
        descs = [d.get('ImageDescription', '') for d in ifds]
        if all(x == descs[0] for x in descs):
            descs = descs[0:1]
        return data, descs
    return data


def _ifd_bytes(width, length, bits, sample_format, desc_len, desc_offset, strip_offset, strip_bytes, next_ifd):
    entries = [  # (tag, type, count, value) in ascending tag order
        (254, 4, 1, 0), (256, 4, 1, width), (257, 4, 1, length), (258, 3, 1, bits),
        (262, 3, 1, 1), (270, 2, desc_len, desc_offset), (273, 4, 1, strip_offset),
        (277, 3, 1, 1), (278, 3, 1, length), (279, 4, 1, strip_bytes), (339, 3, 1, sample_format)]
    out = struct.pack('<H', len(entries))
    for tag, ftype, count, value in entries:
        out += struct.pack('<HHI', tag, ftype, count)
        out += struct.pack('<HH', value, 0) if ftype == 3 else struct.pack('<I', value)
    return out + struct.pack('<I', next_ifd)


def array_to_tif(x, filename, slices=None, channels=None, verbose=False, coerce_64bit_to_32bit=True):
    """Save a 1-, 2- or 3-D numpy array as an ImageJ-readable TIF stack.  64-bit
    data is written as 32-bit unless coerce_64bit_to_32bit is False."""
    x = np.asarray(x)
    if x.ndim == 1:
        x = x.reshape(1, 1, -1)
    elif x.ndim == 2:
        x = x.reshape((1,) + x.shape)
    assert x.ndim == 3
    dtype = x.dtype
    if coerce_64bit_to_32bit and dtype in (np.float64, np.int64, np.uint64):
        dtype = np.dtype({'f': 'float32', 'i': 'int32', 'u': 'uint32'}[dtype.kind])
    if dtype.kind not in _SAMPLE_FORMAT:
        raise UserWarning("Unsupported data format: " + str(dtype))
    n, length, width = x.shape
    bits = dtype.itemsize * 8
    if slices is not None and channels is not None:
        assert slices * channels == n
        desc = ('ImageJ=1.48e\nimages=%i\nchannels=%i\nslices=%i\nhyperstack=true\nmode=grayscale\n'
                'loop=false\nmin=%0.3f\nmax=%0.3f\n\x00' % (n, channels, slices, x.min(), x.max()))
    else:
        desc = 'ImageJ=1.48e\nimages=%i\nslices=%i\nloop=false\nmin=%0.3f\nmax=%0.3f\n\x00' % (
            n, n, x.min(), x.max())
    desc = desc.encode('ascii')
    ifd_size = len(_ifd_bytes(0, 0, 0, 0, 0, 0, 0, 0, 0))
    strip_bytes = length * width * dtype.itemsize
    desc_offset = 8 + ifd_size
    data_offset = desc_offset + len(desc)
    tail_offset = data_offset + n * strip_bytes          # the remaining IFDs follow the pixel data
    with open(filename, 'wb') as f:
        f.write(b'II*\x00' + struct.pack('<I', 8))
        f.write(_ifd_bytes(width, length, bits, _SAMPLE_FORMAT[dtype.kind], len(desc), desc_offset,
                           data_offset, strip_bytes, tail_offset if n > 1 else 0))
        f.write(desc)
        for z in range(n):                                   # one image at a time keeps the peak memory low
            f.write(np.ascontiguousarray(x[z], dtype=dtype.newbyteorder('<')).tobytes())
        for z in range(1, n):
            nxt = tail_offset + z * ifd_size if z < n - 1 else 0
            f.write(_ifd_bytes(width, length, bits, _SAMPLE_FORMAT[dtype.kind], len(desc), desc_offset,
                               data_offset + z * strip_bytes, strip_bytes, nxt))
    if verbose:
        print("Wrote", filename, x.shape, dtype)
    return None
