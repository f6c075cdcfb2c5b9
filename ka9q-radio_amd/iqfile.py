"""iqrecord / iqplay on-disk format: a headerless stream of interleaved int16 I/Q samples, little
endian ("s16le", iqrecord.c:268-272), with the metadata in extended file attributes named user.<key> on Linux
(attr.c:21-78): samplerate, channels, ssrc (hex), sampleformat, frequency, source_timestamp, source, multicast,
unixstarttime (iqrecord.c:262-292).  iqplay reads samplerate, frequency and source_timestamp (falling back to
unixstarttime) back (iqplay.c:188-203) and streams the file in blocks (iqplay.c:248-262).

Host-side file handling only; the samples go to the device through kh_bank_write_input with payload KH_PT_IQ16LE (the
same bytes IQ_PT carries after its status header, radio.c:137-139,159-166).  Where the file system refuses user
attributes, a sidecar file <path>.attrs with one key=value line per attribute stands in."""
import os

import numpy as np

_KEYS = ("samplerate", "channels", "ssrc", "sampleformat", "frequency", "source_timestamp", "source", "multicast",
         "unixstarttime")


def _set(path, key, value):
    try:
        os.setxattr(path, "user." + key, value.encode())  # attrprintf, attr.c:58-78
        return True
    except OSError:
        return False


def write(path, iq_int16, samplerate, frequency, ssrc=0, source_timestamp=None, unixstarttime=None, source="", multicast=""):
    """iq_int16: interleaved I, Q int16 values (any shape, flattened).  Attribute formats follow iqrecord.c:266-291."""
    data = np.ascontiguousarray(iq_int16, np.int16).reshape(-1)
    assert data.size % 2 == 0
    data.astype("<i2").tofile(path)
    attrs = {"samplerate": "%lu" % int(samplerate), "channels": "%d" % 2, "ssrc": "%x" % int(ssrc), "sampleformat": "s16le",
             "frequency": "%.3f" % float(frequency)}
    if source_timestamp is not None:
        attrs["source_timestamp"] = "%d" % int(source_timestamp)
    if unixstarttime is not None:
        attrs["unixstarttime"] = "%.6f" % float(unixstarttime)
    if source:
        attrs["source"] = source
    if multicast:
        attrs["multicast"] = multicast
    if not all([_set(path, k, v) for k, v in attrs.items()]):
        with open(path + ".attrs", "w") as f:
            for k, v in attrs.items():
                f.write(f"{k}={v}\n")
    return attrs


def read_attrs(path):
    out = {}
    for k in _KEYS:
        try:
            out[k] = os.getxattr(path, "user." + k).decode()  # attrscanf, attr.c:21-55
        except OSError:
            pass
    if os.path.exists(path + ".attrs"):
        for line in open(path + ".attrs"):
            k, _, v = line.rstrip("\n").partition("=")
            out.setdefault(k, v)
    return out


UNIX_EPOCH, GPS_UTC_OFFSET = 315964800, 18  # misc: seconds between the UNIX and GPS epochs; leap seconds


def open_iq(path, default_samprate=192000, default_frequency=0.0):
    """-> (memory-mapped int16 array of shape (nsamples, 2), metadata as iqplay.c:188-203 derives it)"""
    a = read_attrs(path)
    meta = {"samprate": int(a.get("samplerate", default_samprate)), "frequency": float(a.get("frequency", default_frequency)),
            "ssrc": int(a["ssrc"], 16) if "ssrc" in a else 0, "sampleformat": a.get("sampleformat", "s16le")}
    if "source_timestamp" in a:
        meta["timestamp"] = int(a["source_timestamp"])
    elif "unixstarttime" in a:  # decimal seconds from the UNIX epoch -> integer nanoseconds from the GPS epoch
        meta["timestamp"] = int((float(a["unixstarttime"]) - UNIX_EPOCH + GPS_UTC_OFFSET) * 1000000000)
    else:
        meta["timestamp"] = 0
    n = os.path.getsize(path) // 4
    data = np.memmap(path, dtype="<i2", mode="r", shape=(n, 2)) if n else np.zeros((0, 2), "<i2")
    return data, meta


def blocks(data, blocksize):
    """iqplay.c:248-262: whole blocks of `blocksize` samples with the power iqplay reports; a short tail is dropped."""
    for k in range(data.shape[0] // blocksize):
        seg = np.asarray(data[k * blocksize:(k + 1) * blocksize])
        v = seg.astype(np.float32)
        yield seg, float((v * v).sum() / (32767.0 * 32767.0 * blocksize))


def replay(path, bank, input_index, nblocks_per_launch, on_launch):
    """Feeds a bank created with payload KH_PT_IQ16LE and the file's sample rate, L samples per block, and calls
    on_launch(first_block, nblocks) after every kh_bank_execute."""
    data, meta = open_iq(path)
    L = bank.L
    nb = data.shape[0] // L
    done = 0
    while done < nb:
        T = min(nblocks_per_launch, nb - done)
        bank.write_input(input_index, np.ascontiguousarray(data[done * L:(done + T) * L]))
        bank.execute(T)
        on_launch(done, T)
        done += T
    return meta, done
