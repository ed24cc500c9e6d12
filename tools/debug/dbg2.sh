python tools/debug/victim_stft.py 16 fp16 3 128
python tools/debug/victim_stft.py 16 fp32 3 128
python tools/debug/victim_stft.py 128 bf16 2 128
python tools/debug/victim_stft.py 16 fp16 3 128 0,0,0
python tools/debug/victim_stft.py 16 fp16 3 128 1,1,0
