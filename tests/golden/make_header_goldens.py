"""Generate golden bytes for the MGARD-X self-describing header (proto3 `mgard.pb.Header`).

Run in the build container only (needs /root/reference and the protoc bundled with torch):

    PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION=python python tests/golden/make_header_goldens.py

It compiles the reference's schema (src/mgard.proto) with protoc into a temporary directory,
fills `Header` messages the way MetadataBase::Serialize does (reference
src/mgard-x/Metadata/Metadata.cpp:249-462, including its quirk: `file_format_version` is only
touched, never assigned, and `mgard_version` ends up holding the FILE format version 1.0.0),
serialises them with the stock Python protobuf runtime and writes the bytes -- data, not code --
to tests/golden/header_goldens.json. The repository's hand-written encoder/decoder
(mgard_amd/csrc/format.hpp) is tested against these bytes; nothing from the reference travels.
"""
import json
import os
import subprocess
import sys
import tempfile
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
PROTO = "/root/reference/src/mgard.proto"
PROTOC = "/usr/local/lib/python3.10/dist-packages/torch/bin/protoc"

CASES = [
    dict(name="f32_3d_rel_inf_huffman", dtype="f32", shape=[512, 512, 512], mode="REL", s="inf",
         norm=1.5008636713027954, tol=1e-3, dd=None, lossless="X_HUFFMAN", dict_size=8192,
         block_size=20480, reorder=0, coords=None),
    dict(name="f64_3d_abs_s0_zstd_maxdim", dtype="f64", shape=[64, 512, 512], mode="ABS", s=0.0,
         norm=0.0, tol=2.5e-4, dd=("MAX_DIMENSION", 1, 256), lossless="X_HUFFMAN_ZSTD",
         dict_size=8192, block_size=20480, reorder=0, coords=None),
    dict(name="f32_4d_rel_s1_block", dtype="f32", shape=[8, 100, 36, 260], mode="REL", s=1.0,
         norm=0.731, tol=1e-2, dd=("BLOCK", 0, 64), lossless="X_HUFFMAN_ZSTD", dict_size=4096,
         block_size=10240, reorder=1, coords=None),
    dict(name="f64_2d_nonuniform", dtype="f64", shape=[5, 7], mode="REL", s=-1.0, norm=3.25, tol=1e-4,
         dd=None, lossless="X_HUFFMAN", dict_size=8192, block_size=20480, reorder=0,
         coords=[[0.0, 0.1, 0.35, 0.7, 1.0], [0.0, 0.2, 0.3, 0.45, 0.5, 0.9, 1.0]]),
    dict(name="f32_1d_cpu_lossless", dtype="f32", shape=[1048576], mode="ABS", s="inf", norm=0.0,
         tol=1e-3, dd=None, lossless="CPU_HUFFMAN_ZSTD", dict_size=0, block_size=0, reorder=0,
         coords=None),
]


def build(pb, c):
    h = pb.Header()
    # Metadata.cpp:252-272 (the quirk)
    h.mgard_version.major_ = 1
    h.mgard_version.minor_ = 0
    h.mgard_version.patch_ = 0
    h.file_format_version.SetInParent()
    h.domain.topology = pb.Domain.CARTESIAN_GRID
    h.domain.cartesian_grid_topology.dimension = len(c["shape"])
    h.domain.cartesian_grid_topology.shape.extend(c["shape"])
    if c["coords"] is None:
        h.domain.geometry = pb.Domain.UNIT_CUBE
    else:
        for cs in c["coords"]:
            h.domain.explicit_cube_geometry.coordinates.extend(cs)
        h.domain.geometry = pb.Domain.EXPLICIT_CUBE
    h.dataset.type = pb.Dataset.DOUBLE if c["dtype"] == "f64" else pb.Dataset.FLOAT
    h.dataset.dimension = 1
    s = float("inf") if c["s"] == "inf" else c["s"]
    if c["mode"] == "ABS":
        h.error_control.mode = pb.ErrorControl.ABSOLUTE
    else:
        h.error_control.mode = pb.ErrorControl.RELATIVE
        h.error_control.norm_of_original_data = c["norm"]
    h.error_control.norm = pb.ErrorControl.L_INFINITY if s == float("inf") else pb.ErrorControl.S_NORM
    h.error_control.s = s
    h.error_control.tolerance = c["tol"]
    if c["dd"] is None:
        h.domain_decomposition.method = pb.DomainDecomposition.NOOP_METHOD
        h.domain_decomposition.decomposition_dimension = 0
        h.domain_decomposition.decomposition_size = c["shape"][0]
    else:
        h.domain_decomposition.method = getattr(pb.DomainDecomposition, c["dd"][0])
        h.domain_decomposition.decomposition_dimension = c["dd"][1]
        h.domain_decomposition.decomposition_size = c["dd"][2]
    h.function_decomposition.transform = pb.FunctionDecomposition.MULTILEVEL_COEFFICIENTS
    h.function_decomposition.hierarchy = pb.FunctionDecomposition.MULTIDIMENSION_WITH_GHOST_NODES
    h.function_decomposition.L_target = 0
    h.quantization.method = pb.Quantization.COEFFICIENTWISE_LINEAR
    h.quantization.bin_widths = pb.Quantization.PER_COEFFICIENT
    h.quantization.type = pb.Quantization.INT64_T
    h.quantization.big_endian = False
    h.bitplane_encoding.method = pb.BitplaneEncoding.NOOP_BITPLANE_ENCODING
    h.encoding.preprocessor = pb.Encoding.SHUFFLE if c["reorder"] else pb.Encoding.NOOP_PREPROCESSOR
    h.encoding.compressor = getattr(pb.Encoding, c["lossless"])
    if c["lossless"] != "CPU_HUFFMAN_ZSTD":
        h.encoding.huffman_dictionary_size = c["dict_size"]
        h.encoding.huffman_block_size = c["block_size"]
    h.device.backend = pb.Device.X_HIP
    return h.SerializeToString()


def main():
    tmp = tempfile.mkdtemp()
    subprocess.check_call([PROTOC, "--python_out=" + tmp, "--proto_path=" + os.path.dirname(PROTO), PROTO])
    sys.path.insert(0, tmp)
    import mgard_pb2 as pb
    out = []
    for c in CASES:
        body = build(pb, c)
        full = b"MGARD" + len(body).to_bytes(8, "little") + zlib.crc32(body).to_bytes(4, "little") + body
        d = dict(c)
        d["header_hex"] = body.hex()
        d["metadata_hex"] = full.hex()
        out.append(d)
    with open(os.path.join(HERE, "header_goldens.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(out), "cases")


if __name__ == "__main__":
    main()
