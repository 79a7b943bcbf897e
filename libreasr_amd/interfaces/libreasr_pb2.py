"""Wire-compatible messages of interfaces/libreasr.proto (reference file, 18 lines):

    package ASR;
    message Audio      { bytes data = 1; int32 sr = 3; }
    message Transcript { string data = 1; }
    service ASR { rpc Transcribe(Audio) returns (Transcript);
                  rpc TranscribeStream(stream Audio) returns (stream Transcript); }

The reference's generated stubs target a 2020 protobuf runtime (`_reflection` API, removed since) and
`grpc_tools` is not installed, so the descriptors are built programmatically; field numbers, types
and the package name are identical, hence the bytes on the wire are identical."""
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

_fd = descriptor_pb2.FileDescriptorProto()
_fd.name = "libreasr_amd/libreasr.proto"
_fd.package = "ASR"
_fd.syntax = "proto3"

_m = _fd.message_type.add()
_m.name = "Audio"
_f = _m.field.add(); _f.name = "data"; _f.number = 1
_f.type = descriptor_pb2.FieldDescriptorProto.TYPE_BYTES; _f.label = descriptor_pb2.FieldDescriptorProto.LABEL_OPTIONAL
_f = _m.field.add(); _f.name = "sr"; _f.number = 3
_f.type = descriptor_pb2.FieldDescriptorProto.TYPE_INT32; _f.label = descriptor_pb2.FieldDescriptorProto.LABEL_OPTIONAL

_m = _fd.message_type.add()
_m.name = "Transcript"
_f = _m.field.add(); _f.name = "data"; _f.number = 1
_f.type = descriptor_pb2.FieldDescriptorProto.TYPE_STRING; _f.label = descriptor_pb2.FieldDescriptorProto.LABEL_OPTIONAL

_s = _fd.service.add()
_s.name = "ASR"
_r = _s.method.add(); _r.name = "Transcribe"; _r.input_type = ".ASR.Audio"; _r.output_type = ".ASR.Transcript"
_r = _s.method.add(); _r.name = "TranscribeStream"; _r.input_type = ".ASR.Audio"; _r.output_type = ".ASR.Transcript"
_r.client_streaming = True; _r.server_streaming = True

_pool = descriptor_pool.DescriptorPool()
DESCRIPTOR = _pool.AddSerializedFile(_fd.SerializeToString())
Audio = message_factory.GetMessageClass(_pool.FindMessageTypeByName("ASR.Audio"))
Transcript = message_factory.GetMessageClass(_pool.FindMessageTypeByName("ASR.Transcript"))
