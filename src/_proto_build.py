"""Builds protobuf message classes from hand-written descriptors (there is no protoc here).
The schema is the reference's, field for field: package `src.proto`, files `src/trajectory.proto`,
`src/ilqr_options.proto`, `src/ilqr_debug.proto`."""
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

_F = descriptor_pb2.FieldDescriptorProto
DOUBLE, INT32, BOOL, MESSAGE = _F.TYPE_DOUBLE, _F.TYPE_INT32, _F.TYPE_BOOL, _F.TYPE_MESSAGE


def build_file(name, messages, deps=()):
    """messages: [(msg_name, [(field_name, number, type, type_name|None, repeated)])]"""
    pool = descriptor_pool.Default()
    try:
        fd = pool.FindFileByName(name)
    except KeyError:
        fdp = descriptor_pb2.FileDescriptorProto()
        fdp.name = name
        fdp.package = "src.proto"
        fdp.syntax = "proto3"
        fdp.dependency.extend(deps)
        for mname, fields in messages:
            m = fdp.message_type.add()
            m.name = mname
            for fname, num, ftype, tname, rep in fields:
                f = m.field.add()
                f.name = fname
                f.number = num
                f.type = ftype
                f.label = _F.LABEL_REPEATED if rep else _F.LABEL_OPTIONAL
                if tname:
                    f.type_name = ".src.proto." + tname
        fd = pool.Add(fdp) if hasattr(pool, "Add") else None
        fd = pool.FindFileByName(name)
    out = {}
    for mname, _ in messages:
        out[mname] = message_factory.GetMessageClass(fd.message_types_by_name[mname])
    return fd, out
