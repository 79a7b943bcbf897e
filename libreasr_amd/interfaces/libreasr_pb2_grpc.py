"""Client stub / servicer base / registration for service ASR.ASR (same method paths as the
reference's interfaces/libreasr_pb2_grpc.py: /ASR.ASR/Transcribe, /ASR.ASR/TranscribeStream)."""
import grpc

from . import libreasr_pb2 as pb


class ASRStub:
    def __init__(self, channel):
        self.Transcribe = channel.unary_unary(
            "/ASR.ASR/Transcribe", request_serializer=pb.Audio.SerializeToString,
            response_deserializer=pb.Transcript.FromString)
        self.TranscribeStream = channel.stream_stream(
            "/ASR.ASR/TranscribeStream", request_serializer=pb.Audio.SerializeToString,
            response_deserializer=pb.Transcript.FromString)


class ASRServicer:
    def Transcribe(self, request, context):
        context.set_code(grpc.StatusCode.UNIMPLEMENTED)
        raise NotImplementedError("Method not implemented!")

    def TranscribeStream(self, request_iterator, context):
        context.set_code(grpc.StatusCode.UNIMPLEMENTED)
        raise NotImplementedError("Method not implemented!")


def add_ASRServicer_to_server(servicer, server):
    handlers = {
        "Transcribe": grpc.unary_unary_rpc_method_handler(
            servicer.Transcribe, request_deserializer=pb.Audio.FromString,
            response_serializer=pb.Transcript.SerializeToString),
        "TranscribeStream": grpc.stream_stream_rpc_method_handler(
            servicer.TranscribeStream, request_deserializer=pb.Audio.FromString,
            response_serializer=pb.Transcript.SerializeToString),
    }
    server.add_generic_rpc_handlers((grpc.method_handlers_generic_handler("ASR.ASR", handlers),))
