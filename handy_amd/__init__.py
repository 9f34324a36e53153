"""handy_amd: MI355X-native Whisper engine behind the whisper.h C ABI that Handy reaches through
whisper-rs (reference call site: src-tauri/src/managers/transcription.rs:391-416)."""
__version__ = "0.1.0"
