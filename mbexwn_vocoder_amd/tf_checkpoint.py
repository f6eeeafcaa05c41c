"""TensorFlow-2 checkpoint (``weights.tf.index`` + ``weights.tf.data-?????-of-?????``) reader without TensorFlow.

The reference restores its pretrained models with ``model.load_weights(model_dir/"weights.tf")``
(MBExWN_NVoc/mel_inverter.py:206-210); the files are TensorFlow "tensor bundles".  TensorFlow is a third-party
dependency that is not in the reference tree and not installable here, so this module restates the published formats
(tensorflow/core/util/tensor_bundle, tensorflow/core/lib/io/{format,block,table}, leveldb table format,
tensorflow/core/protobuf/{tensor_bundle,trackable_object_graph}.proto):

* ``.index``: a leveldb-style sorted string table.  Blocks of prefix-compressed entries
  ``varint shared | varint unshared | varint value_len | key suffix | value`` followed by a restart array and its
  length (uint32 LE); after each block 1 byte compression type (0 none, 1 snappy) + 4 bytes masked CRC32C;
  48-byte footer = metaindex handle, index handle (varint offset, varint size), padding, magic 0xdb4775248b80fb57.
  Key "" holds a ``BundleHeaderProto``; every other key is a checkpoint key with a ``BundleEntryProto``
  (dtype, shape, shard_id, offset, size, crc32c).
* ``.data-*``: the raw little-endian tensor bytes at (shard, offset, size).  A scalar DT_STRING tensor is stored as
  varint length(s), a 4-byte checksum of the lengths, then the bytes.
* key ``_CHECKPOINTABLE_OBJECT_GRAPH``: serialized ``TrackableObjectGraph``; its nodes carry, per variable, the
  variable's ``full_name`` (e.g. ``PulsPar_Layer_0/kernel``) and its ``checkpoint_key``.

**Not verified against a file written by TensorFlow** (none exists in this environment):
the tests round-trip through :func:`write_checkpoint`, which follows the same published formats, and check the
building blocks against known answers (CRC32C, varints, snappy, prefix-compressed multi-block tables).
"""
import os
import re
import struct

import numpy as np

_TABLE_MAGIC = 0xDB4775248B80FB57
_MASK_DELTA = 0xA282EAD8
OBJECT_GRAPH_KEY = "_CHECKPOINTABLE_OBJECT_GRAPH"

# tensorflow/core/framework/types.proto
_DTYPES = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 4: np.dtype("u1"), 5: np.dtype("<i2"),
           6: np.dtype("i1"), 9: np.dtype("<i8"), 10: np.dtype("bool"), 17: np.dtype("<u2"), 19: np.dtype("<f2"),
           22: np.dtype("<u4"), 23: np.dtype("<u8")}
_DT_STRING = 7
_DTYPE_CODES = {np.dtype(vv).newbyteorder("="): kk for kk, vv in _DTYPES.items()}


# ------------------------------------------------------------------------------------------------------
# CRC32C (Castagnoli), masked as leveldb / TensorFlow store it
# ------------------------------------------------------------------------------------------------------
def _make_crc_table():
    table = np.zeros(256, dtype=np.uint32)
    for ii in range(256):
        crc = ii
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
        table[ii] = crc
    return table


_CRC_TABLE = _make_crc_table()


def crc32c(data, crc=0):
    crc ^= 0xFFFFFFFF
    table = _CRC_TABLE
    for byte in bytes(data):
        crc = int(table[(crc ^ byte) & 0xFF]) ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def mask_crc(crc):
    return (((crc >> 15) | (crc << 17)) + _MASK_DELTA) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------------
# varints and the few protobuf messages involved (wire format only)
# ------------------------------------------------------------------------------------------------------
def read_varint(buf, pos):
    result, shift = 0, 0
    while True:
        byte = buf[pos]
        pos += 1
        result |= (byte & 0x7F) << shift
        if not byte & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError("malformed varint")


def write_varint(value):
    out = bytearray()
    value &= (1 << 64) - 1
    while True:
        byte = value & 0x7F
        value >>= 7
        if value:
            out.append(byte | 0x80)
        else:
            out.append(byte)
            return bytes(out)


def parse_message(buf):
    """protobuf wire format -> {field number: [values]} (varint -> int, 64-bit/32-bit -> bytes, length-delimited -> bytes)."""
    fields, pos = {}, 0
    buf = bytes(buf)
    while pos < len(buf):
        tag, pos = read_varint(buf, pos)
        number, wire = tag >> 3, tag & 7
        if wire == 0:
            value, pos = read_varint(buf, pos)
        elif wire == 1:
            value, pos = buf[pos:pos + 8], pos + 8
        elif wire == 2:
            size, pos = read_varint(buf, pos)
            value, pos = buf[pos:pos + size], pos + size
        elif wire == 5:
            value, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wire}")
        fields.setdefault(number, []).append(value)
    return fields


def _field(number, wire, payload):
    tag = write_varint((number << 3) | wire)
    if wire == 0:
        return tag + write_varint(payload)
    if wire == 2:
        return tag + write_varint(len(payload)) + payload
    if wire == 5:
        return tag + struct.pack("<I", payload)
    raise ValueError(wire)


def _signed(value):
    return value - (1 << 64) if value >= (1 << 63) else value


def parse_bundle_entry(buf):
    """BundleEntryProto -> dict(dtype, shape, shard_id, offset, size, crc32c, sliced)."""
    msg = parse_message(buf)
    shape = []
    if 2 in msg:
        for dim in parse_message(msg[2][0]).get(2, []):
            shape.append(_signed(parse_message(dim).get(1, [0])[0]))
    return {"dtype": msg.get(1, [0])[0], "shape": tuple(shape), "shard_id": msg.get(3, [0])[0],
            "offset": msg.get(4, [0])[0], "size": msg.get(5, [0])[0],
            "crc32c": struct.unpack("<I", msg[6][0])[0] if 6 in msg else None, "sliced": 7 in msg}


def encode_bundle_entry(dtype_code, shape, shard_id, offset, size, crc):
    dims = b"".join(_field(2, 2, _field(1, 0, int(dd))) for dd in shape)
    return (_field(1, 0, dtype_code) + _field(2, 2, dims) + _field(3, 0, shard_id) + _field(4, 0, offset) +
            _field(5, 0, size) + _field(6, 5, crc))


def parse_object_graph(buf):
    """TrackableObjectGraph -> [(full_name, checkpoint_key, attribute name)] over all nodes."""
    out = []
    for node in parse_message(buf).get(1, []):
        for attr in parse_message(node).get(2, []):
            fields = parse_message(attr)
            name = fields.get(1, [b""])[0].decode()
            full_name = fields.get(2, [b""])[0].decode()
            key = fields.get(3, [b""])[0].decode()
            out.append((full_name, key, name))
    return out


# ------------------------------------------------------------------------------------------------------
# snappy (raw format) -- TensorFlow writes the bundle index uncompressed, other writers may not
# ------------------------------------------------------------------------------------------------------
def snappy_decompress(buf):
    buf = bytes(buf)
    total, pos = read_varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:
            size = tag >> 2
            if size >= 60:
                nbytes = size - 59
                size = int.from_bytes(buf[pos:pos + nbytes], "little")
                pos += nbytes
            size += 1
            out += buf[pos:pos + size]
            pos += size
            continue
        if kind == 1:
            size = ((tag >> 2) & 7) + 4
            offset = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            size = (tag >> 2) + 1
            offset = int.from_bytes(buf[pos:pos + 2], "little")
            pos += 2
        else:
            size = (tag >> 2) + 1
            offset = int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        if offset == 0 or offset > len(out):
            raise ValueError("malformed snappy stream")
        for _ in range(size):           # copies may overlap their own output
            out.append(out[-offset])
    if len(out) != total:
        raise ValueError("snappy length mismatch")
    return bytes(out)


# ------------------------------------------------------------------------------------------------------
# sorted string table
# ------------------------------------------------------------------------------------------------------
def _read_block(buf, offset, size, verify):
    contents = buf[offset:offset + size]
    kind = buf[offset + size]
    if verify:
        stored = struct.unpack("<I", buf[offset + size + 1:offset + size + 5])[0]
        if mask_crc(crc32c(buf[offset:offset + size + 1])) != stored:
            raise ValueError(f"checksum mismatch in table block at {offset}")
    if kind == 1:
        contents = snappy_decompress(contents)
    elif kind != 0:
        raise ValueError(f"unknown block compression type {kind}")
    return contents


def _block_entries(block):
    n_restarts = struct.unpack("<I", block[-4:])[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = read_varint(block, pos)
        unshared, pos = read_varint(block, pos)
        vlen, pos = read_varint(block, pos)
        key = key[:shared] + block[pos:pos + unshared]
        pos += unshared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_table(path, verify=True):
    """All (key, value) pairs of a leveldb-format table file, in key order."""
    with open(path, "rb") as fi:
        buf = fi.read()
    if len(buf) < 48 or struct.unpack("<Q", buf[-8:])[0] != _TABLE_MAGIC:
        raise ValueError(f"{path} is not a TensorFlow checkpoint index (bad table magic)")
    footer = buf[-48:]
    _, pos = read_varint(footer, 0)          # metaindex handle (unused)
    _, pos = read_varint(footer, pos)
    index_offset, pos = read_varint(footer, pos)
    index_size, pos = read_varint(footer, pos)
    out = []
    for _, handle in _block_entries(_read_block(buf, index_offset, index_size, verify)):
        offset, hp = read_varint(handle, 0)
        size, _ = read_varint(handle, hp)
        out.extend(_block_entries(_read_block(buf, offset, size, verify)))
    return out


def _build_block(entries, restart_interval=16):
    out, restarts, last = bytearray(), [], b""
    for ii, (key, value) in enumerate(entries):
        shared = 0
        if ii % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(key), len(last)) and key[shared] == last[shared]:
                shared += 1
        out += write_varint(shared) + write_varint(len(key) - shared) + write_varint(len(value))
        out += key[shared:] + value
        last = key
    if not restarts:
        restarts = [0]
    for rr in restarts:
        out += struct.pack("<I", rr)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def write_table(path, items, block_entries=64):
    """items: iterable of (key bytes, value bytes); written sorted, uncompressed, with block checksums."""
    items = sorted(items)
    blob, index = bytearray(), []

    def emit(block):
        handle = write_varint(len(blob)) + write_varint(len(block))
        blob.extend(block)
        blob.extend(b"\x00" + struct.pack("<I", mask_crc(crc32c(block + b"\x00"))))
        return handle

    for start in range(0, max(len(items), 1), block_entries):
        chunk = items[start:start + block_entries]
        handle = emit(_build_block(chunk))
        index.append((chunk[-1][0] if chunk else b"", handle))
    meta_handle = emit(_build_block([]))
    index_handle = emit(_build_block(index, restart_interval=1))
    footer = meta_handle + index_handle
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", _TABLE_MAGIC)
    with open(path, "wb") as fo:
        fo.write(bytes(blob) + footer)


# ------------------------------------------------------------------------------------------------------
# tensor bundle
# ------------------------------------------------------------------------------------------------------
def _shard_path(prefix, shard, n_shards):
    return f"{prefix}.data-{shard:05d}-of-{n_shards:05d}"


class CheckpointReader(object):
    """Lazy reader of one checkpoint prefix (``.../weights.tf``)."""

    def __init__(self, prefix, verify_index=True):
        self.prefix = prefix
        index_path = prefix + ".index"
        if not os.path.exists(index_path):
            raise FileNotFoundError(index_path)
        self.entries = {}
        self.n_shards = 1
        for key, value in read_table(index_path, verify=verify_index):
            if key == b"":
                header = parse_message(value)
                self.n_shards = header.get(1, [1])[0]
                if header.get(2, [0])[0] != 0:
                    raise NotImplementedError("big-endian tensor bundle")
            else:
                self.entries[key.decode()] = parse_bundle_entry(value)
        self._shards = {}

    def keys(self):
        return sorted(self.entries)

    def _bytes(self, entry):
        shard = entry["shard_id"]
        if shard not in self._shards:
            self._shards[shard] = np.memmap(_shard_path(self.prefix, shard, self.n_shards), dtype=np.uint8, mode="r")
        return self._shards[shard][entry["offset"]:entry["offset"] + entry["size"]]

    def get_bytes(self, key):
        """Scalar DT_STRING entry (the object graph)."""
        entry = self.entries[key]
        if entry["dtype"] != _DT_STRING or int(np.prod(entry["shape"], dtype=np.int64)) != 1:
            raise ValueError(f"{key} is not a scalar string tensor")
        raw = bytes(self._bytes(entry))
        length, pos = read_varint(raw, 0)
        return raw[pos + 4:pos + 4 + length]          # 4 bytes: checksum of the length prefix

    def get_tensor(self, key, verify=False):
        entry = self.entries[key]
        if entry["sliced"]:
            raise NotImplementedError(f"{key}: partitioned (sliced) variables are not supported")
        if entry["dtype"] not in _DTYPES:
            raise NotImplementedError(f"{key}: unsupported dtype code {entry['dtype']}")
        raw = self._bytes(entry)
        if verify and entry["crc32c"] is not None and mask_crc(crc32c(raw)) != entry["crc32c"]:
            raise ValueError(f"checksum mismatch in tensor {key}")
        return np.frombuffer(bytes(raw), dtype=_DTYPES[entry["dtype"]]).reshape(entry["shape"]).copy()

    def variables(self):
        """{variable full_name: checkpoint key} from the object graph (falls back to the keys themselves)."""
        out = {}
        if OBJECT_GRAPH_KEY in self.entries:
            for full_name, key, attr in parse_object_graph(self.get_bytes(OBJECT_GRAPH_KEY)):
                if key in self.entries and attr == "VARIABLE_VALUE":
                    out[full_name or key] = key
        for key in self.entries:
            if key != OBJECT_GRAPH_KEY and key not in out.values():
                out[key.replace("/.ATTRIBUTES/VARIABLE_VALUE", "")] = key
        return out


def write_checkpoint(prefix, variables):
    """Write {variable full_name: array} as a one-shard TF2 object-based checkpoint (flat object graph: one child of
    the root per variable).  Used by the tests and to hand converted weights back to TensorFlow users."""
    data = bytearray()
    items = [(b"", _field(1, 0, 1) + _field(2, 0, 0) + _field(3, 2, _field(1, 0, 1)))]
    nodes = []
    children = b""
    for ii, (name, value) in enumerate(sorted(variables.items())):
        arr = np.ascontiguousarray(value)
        code = _DTYPE_CODES.get(arr.dtype.newbyteorder("="))
        if code is None:
            raise NotImplementedError(f"dtype {arr.dtype}")
        raw = arr.astype(arr.dtype.newbyteorder("<")).tobytes()
        key = f"v{ii}/.ATTRIBUTES/VARIABLE_VALUE"
        items.append((key.encode(), encode_bundle_entry(code, arr.shape, 0, len(data), len(raw), mask_crc(crc32c(raw)))))
        data += raw
        children += _field(1, 2, _field(1, 0, ii + 1) + _field(2, 2, f"v{ii}".encode()))
        attr = _field(1, 2, b"VARIABLE_VALUE") + _field(2, 2, name.encode()) + _field(3, 2, key.encode())
        nodes.append(_field(2, 2, attr))
    graph = _field(1, 2, children) + b"".join(_field(1, 2, nn) for nn in nodes)
    length = write_varint(len(graph))
    raw = length + struct.pack("<I", mask_crc(crc32c(length))) + graph
    items.append((OBJECT_GRAPH_KEY.encode(),
                  encode_bundle_entry(_DT_STRING, (), 0, len(data), len(raw), mask_crc(crc32c(graph)))))
    data += raw
    write_table(prefix + ".index", items)
    with open(_shard_path(prefix, 0, 1), "wb") as fo:
        fo.write(bytes(data))


# ------------------------------------------------------------------------------------------------------
# reference variable names -> the engine's weight names
# ------------------------------------------------------------------------------------------------------
# Keras names a variable <layer name>/<weight name>.  The reference wraps every convolution in
# TF2C_Conv1DWeightNorm: the inner Conv1D carries the layer's name and owns `kernel` (the direction v) and `bias`, the
# wrapper is called <name>_base and owns `g` (tf2_components/layers/conv_layers.py:52-112).  The wrapper builds the inner
# layer from inside its own build() (conv_layers.py:72-75, tf2c_base_layer.py:34-37), so depending on the Keras name
# scope in force the inner variables appear as `<name>/kernel` or as `<name>_base/kernel` (and `g` as `<name>_base/g`
# or `<name>/g`): both spellings are accepted.  PReLU slopes are `alpha`.  NOT verified against a TensorFlow-written
# checkpoint (none exists in this environment): treat a model loaded this way as unverified until one has been tested.
# Layer names: sub-nets custom_pulsed_generator.py:100-148, WaveNet custom_AE_layers.py:182-257, post-net :490-493.
_WAVENET_LAYERS = {"start": "wn.start", "end": "wn.end", "cond_": "wn.cond"}


def _engine_layer_name(layer):
    if layer in _WAVENET_LAYERS:
        return _WAVENET_LAYERS[layer]
    if re.fullmatch(r"(conv1D|res_skip)_\d+(g\d+)?", layer) or re.fullmatch(r"precond_\d+", layer):
        return "wn." + layer
    if layer.endswith("_PaNMPulseWaveNet_Post"):
        return "post"
    up = re.fullmatch(r"PP_waveNetBlock_ups\d+_(\d+)_WNBlock_UP_\d+", layer)     # custom_AE_layers.py:519-524
    if up:
        return "up" + up.group(1)
    if re.fullmatch(r"\w+_(Layer_(\d+|final)|ActLayer_\d+)", layer):
        return layer
    return None


def map_reference_variables(named_arrays):
    """{TensorFlow variable name: array} -> the raw weight dict of weights.py (<layer>.v / .g / .bias, <act>.alpha).

    Matching is by the last two components of the variable name (`<layer>/kernel`, `<layer>_base/g`, `<layer>/bias`,
    `.../<act layer>/.../alpha`), whatever the enclosing model scopes are.  Returns (weights, unmatched names)."""
    weights, unmatched = {}, []
    for full_name, arr in named_arrays.items():
        parts = full_name.split(":")[0].split("/")
        leaf = parts[-1]
        target = None
        owner = parts[-2] if len(parts) >= 2 else ""
        if owner.endswith("_base"):
            owner = owner[:-len("_base")]
        if leaf in ("kernel", "bias", "g") and owner:
            layer = _engine_layer_name(owner)
            if layer and layer.startswith("wn."):
                # the layers of the WaveNet blocks behind the first one: the block index is in the enclosing scope
                # "PP_waveNetBlock_ups<u>_<i>" (reference custom_pulsed_generator.py:487)
                for comp in parts[:-2]:
                    blk = re.match(r"PP_waveNetBlock_ups\d+_(\d+)", comp)
                    if blk and int(blk.group(1)) > 0:
                        layer = f"wn{int(blk.group(1))}." + layer[3:]
                        break
            if layer:
                target = layer + "." + ("v" if leaf == "kernel" else leaf)
        elif leaf == "alpha":
            for comp in reversed(parts[:-1]):
                layer = _engine_layer_name(comp)
                if layer and "ActLayer" in layer:
                    target = f"{layer}.alpha"
                    break
        if target is None or target in weights:
            unmatched.append(full_name)
        else:
            weights[target] = np.asarray(arr, dtype=np.float32)
    return weights, unmatched


def to_reference_variables(raw, model_scope="mb_ex_wn", config=None):
    """Inverse of :func:`map_reference_variables`: the engine's raw weights under the variable names the reference's
    Keras model gives them (enclosing scopes are a guess and irrelevant to the reader).  WaveNet block ``i`` is named
    ``PP_waveNetBlock_ups{ups_i}_{i}`` (reference custom_pulsed_generator.py:487), its WaveNet ``<block>_WNBlock_WN``
    (custom_AE_layers.py:516) and its up-sampling convolution -- which only exists when ``ups_i > 1`` --
    ``<block>_WNBlock_UP_{ups_i}`` (custom_AE_layers.py:519-526): ``config`` supplies the factors
    (``pp_mod_subnet_upsampling_factors``; without it a single block without upsampling is assumed)."""
    ups = [1]
    if config is not None:
        from .config import ModelDims
        ups = list(ModelDims(config).wn_block_ups)
    out = {}
    inverse = {vv: kk for kk, vv in _WAVENET_LAYERS.items()}

    def block_name(idx):
        if idx >= len(ups):
            raise ValueError(f"weights of WaveNet block {idx}, but the configuration has {len(ups)} block(s): pass config")
        return f"PP_waveNetBlock_ups{ups[idx]}_{idx}"

    for name, arr in raw.items():
        layer, kind = name.rsplit(".", 1)
        blk = re.match(r"wn(\d*)\.(.*)", layer)
        up = re.fullmatch(r"up(\d+)", layer)
        if blk:
            idx, inner = int(blk.group(1) or 0), "wn." + blk.group(2)
            ref = inverse.get(inner, blk.group(2))
            scope = f"{model_scope}/{block_name(idx)}/{block_name(idx)}_WNBlock_WN/"
        elif up:
            idx = int(up.group(1))
            if idx >= len(ups) or ups[idx] <= 1:
                raise ValueError(f"{name}: block {idx} has no up-sampling convolution in this configuration (pass config)")
            ref = f"{block_name(idx)}_WNBlock_UP_{ups[idx]}"
            scope = f"{model_scope}/{block_name(idx)}/"
        elif layer == "post":
            ref, scope = f"{model_scope}_PaNMPulseWaveNet_Post", f"{model_scope}/"
        else:
            ref, scope = layer, f"{model_scope}/"
        if kind == "v":
            out[f"{scope}{ref}/kernel"] = np.asarray(arr)
        elif kind == "bias":
            out[f"{scope}{ref}/bias"] = np.asarray(arr)
        elif kind == "g":
            out[f"{scope}{ref}_base/g"] = np.asarray(arr)
        elif kind == "alpha":
            out[f"{scope}{ref}/p_re_lu/alpha"] = np.asarray(arr).reshape(1, -1)   # Keras PReLU(shared_axes=[1])
        else:
            raise ValueError(f"unknown weight kind in {name}")
    return out


def _uses_weight_norm(config, layer):
    """True when the layer's checkpoint holds a gain g.  WaveNet layers follow pp_mod_subnet.use_weight_norm /
    use_equalized_lr (reference custom_AE_layers.py:123 default False; conv_layers.py:79-119: either option adds g); the
    sub-nets and the post-net are always weight-normed (custom_pulsed_generator.py:100-148, 490-493)."""
    if re.match(r"wn\d*\.", layer):
        wn = config["mbexwn_config"]["pp_mod_subnet"]
        return bool(wn.get("use_weight_norm", False)) or bool(wn.get("use_equalized_lr", False))
    return True


def load_reference_checkpoint(prefix, config=None):
    """Read ``prefix``(.index/.data-*) and return the engine's raw weight dict; with ``config`` the result is checked
    against weights.layer_table (missing tensors / wrong shapes raise)."""
    reader = CheckpointReader(prefix)
    named = {name: reader.get_tensor(key) for name, key in reader.variables().items()
             if reader.entries[key]["dtype"] in _DTYPES}
    weights, unmatched = map_reference_variables(named)
    if config is not None:
        from .weights import layer_table
        convs, prelus = layer_table(config)
        problems = []
        for name, ks, cin, cout in convs:
            for suffix, shape in ((".v", (ks, cin, cout)), (".g", (cout,)), (".bias", (cout,))):
                got = weights.get(name + suffix)
                if got is None:
                    # a layer built with use_weight_norm=False has no g: its kernel is the weight (weights.fold_weights)
                    if suffix != ".g" or _uses_weight_norm(config, name):
                        problems.append(f"missing {name}{suffix}")
                elif tuple(got.shape) != shape:
                    problems.append(f"{name}{suffix} has shape {tuple(got.shape)}, expected {shape}")
        for name, channels in prelus:
            got = weights.get(name + ".alpha")
            if got is None:
                problems.append(f"missing {name}.alpha")
            elif got.size != channels:
                problems.append(f"{name}.alpha has {got.size} values, expected {channels}")
            else:
                weights[name + ".alpha"] = got.reshape(channels)
        if problems:
            raise ValueError("checkpoint does not match the model configuration: " + "; ".join(problems[:8]) +
                             (f" ... (+{len(problems) - 8})" if len(problems) > 8 else "") +
                             (f"; unmatched variables: {unmatched[:5]}" if unmatched else ""))
    return weights
