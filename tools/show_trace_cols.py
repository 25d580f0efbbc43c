import csv,sys,glob,os
for f in glob.glob(os.path.join(sys.argv[1],"**","*kernel_trace.csv"),recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Queue_Id"], r["Stream_Id"], r["Kernel_Name"][:20], r["Workgroup_Size_X"], r["Grid_Size_X"])
